"""CPU suite: the engine-1 device code of the NON-MISP models (knapsack, max-cut, MAX2SAT), compiled as the lock-step host
emulation and fed with the product library's own model descriptors.  A decision diagram compiled EXACTLY (width larger than
any layer) is a dynamic programme: its best value is the optimum; restricted / relaxed diagrams at small widths bracket it
(clean.rs:345-381; relaxed >= optimum >= restricted), and an exact relaxed diagram proves it."""
import itertools

import numpy as np
import pytest

import ddo_amd
from tests.conftest import data_path
from tests.emul_binding import ModelEmul

EXACT, RELAXED, RESTRICTED = 0, 1, 2


def bracket(model, optimum, big, widths):
    e = ModelEmul(model, big)
    r = e.compile_root(EXACT, big)[0]
    assert r["status"] == 0 and r["is_exact"] and r["best_value"] == optimum
    for w in widths:
        lo = e.compile_root(RESTRICTED, w)[0]
        hi = e.compile_root(RELAXED, w)[0]
        assert lo["status"] == 0 and hi["status"] == 0
        if lo["best_value"] is not None:
            assert lo["best_value"] <= optimum
        assert hi["best_value"] is not None and hi["best_value"] >= optimum
        if hi["is_exact"]:
            assert hi["best_value"] == optimum
        else:                      # the cut-set of an inexact relaxed DD carries upper bounds that still cover the optimum
            assert lo["best_value"] == optimum or (hi["cutset"] and max(u for (_, _, u, _) in hi["cutset"]) >= optimum)


def read_kp(path):
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("c")]
    n, cap = int(rows[0][0]), int(rows[0][1])
    return cap, [int(r[0]) for r in rows[1:1 + n]], [int(r[1]) for r in rows[1:1 + n]]


@pytest.mark.parametrize("name,optimum", [("f3_l-d_kp_4_20", 35), ("f4_l-d_kp_4_11", 23), ("f9_l-d_kp_5_80", 130),
                                          ("f7_l-d_kp_7_50", 107), ("f1_l-d_kp_10_269", 295)])
def test_knapsack_device_code_on_cpu(name, optimum):
    model = ddo_amd.Knapsack.read_instance(data_path("knapsack", name))
    bracket(model, optimum, 1500, [1, 2, 3, 8])


def test_knapsack_readme_on_cpu():
    bracket(ddo_amd.Knapsack.from_items(50, [60, 100, 120], [10, 20, 30]), 220, 64, [1, 2])


def cut_optimum(adj):
    n = adj.shape[0]
    best = -10**9
    for m in range(1 << (n - 1)):
        sides = [1] + [1 if (m >> i) & 1 else -1 for i in range(n - 1)]
        best = max(best, int(sum(adj[a, b] for a in range(n) for b in range(a + 1, n) if sides[a] * sides[b] < 0)))
    return best


@pytest.mark.parametrize("seed,n", [(1, 6), (2, 8), (3, 9), (4, 10)])
def test_mcp_device_code_on_cpu(seed, n):
    rng = np.random.RandomState(seed)
    adj = np.zeros((n, n), dtype=np.int64)
    for a in range(n):
        for b in range(a + 1, n):
            if rng.rand() < 0.6:
                adj[a, b] = adj[b, a] = rng.randint(-6, 9)
    bracket(ddo_amd.Mcp.from_matrix(adj), cut_optimum(adj), 1 << n, [1, 2, 3, 7])


def sat_optimum(n, clauses):
    w = {}
    for a, b, c in clauses:
        w[(min(a, b), max(a, b))] = c
    best = -10**9
    for bits in itertools.product((False, True), repeat=n):
        val = lambda lit: bits[abs(lit) - 1] == (lit > 0)
        best = max(best, sum(c for (a, b), c in w.items() if val(a) or val(b)))
    return best


@pytest.mark.parametrize("name,optimum", [("debug", 24), ("debug2", 13), ("pass", 54), ("tautology", 7), ("unit", 6),
                                          ("negative_wt", 4258)])
def test_max2sat_reference_instances_on_cpu(name, optimum):
    model = ddo_amd.Max2Sat.read_instance(data_path("max2sat", name + ".wcnf"))
    bracket(model, optimum, 1 << model.n if model.n <= 10 else 4096, [1, 2, 3])


@pytest.mark.parametrize("seed,n", [(7, 5), (8, 7), (9, 8)])
def test_max2sat_device_code_on_cpu(seed, n):
    rng = np.random.RandomState(seed)
    clauses = []
    for _ in range(3 * n):
        a = int(rng.randint(1, n + 1)) * (1 if rng.rand() < 0.5 else -1)
        b = int(rng.randint(1, n + 1)) * (1 if rng.rand() < 0.5 else -1)
        clauses.append((a, b, int(rng.randint(-4, 12))))
    bracket(ddo_amd.Max2Sat.from_clauses(n, clauses), sat_optimum(n, clauses), 1 << n, [1, 2, 5])


# ---- per-compile parity of the signed-vector models (MAX2SAT, MCP) ---------------------------------------------------
# The reference ranks these states by sum |benefit| only and leaves ties to hash order; oracle and device share ONE
# deterministic tie-break (packed state words, oracle/models.hpp compare_signed_vectors == lexkey in misp_dd_core.hpp), so
# every compile() of an oracle search can be replayed: values, counters and cut-set multisets must be equal.
from tests.parity_util import diff  # noqa: E402

VEC_CASES = [
    ("max2sat", "pass.wcnf", 2, 0), ("max2sat", "pass.wcnf", 3, 0), ("max2sat", "debug2.wcnf", 1, 0), ("max2sat", "unit.wcnf", 2, 0),
    ("max2sat", "negative_wt.wcnf", 2, 0), ("max2sat", "tautology.wcnf", 1, 0),
    ("max2sat", "frb10-6-1.wcnf", 8, 40), ("max2sat", "frb10-6-2.wcnf", 25, 30), ("max2sat", "frb10-6-3.wcnf", 0, 30),
    ("mcp", "mcp_n30_p0.1_000.mcp", 3, 60), ("mcp", "mcp_n30_p0.1_001.mcp", 10, 60), ("mcp", "mcp_n30_p0.1_004.mcp", 0, 60),
    ("mcp", "mcp_n30_p0.1_007.mcp", 2, 80),
    ("max2sat", "frb15-9-1.wcnf", 6, 14),     # n = 135: 69 state words, the 72-word template (BASELINE config C3's stress instance)
]


@pytest.mark.parametrize("kind,fname,width,max_compiles", VEC_CASES)
def test_vector_models_replay_oracle_trace(oracle, kind, fname, width, max_compiles):
    path = data_path(kind, fname)
    model = (ddo_amd.Max2Sat if kind == "max2sat" else ddo_amd.Mcp).read_instance(path)
    _, recs = oracle.vector_trace(kind, path, width, max_compiles)
    assert recs and len(recs[0]["state"]) == model.ws
    e = ModelEmul(model, max(int(r["width"]) for r in recs))
    relaxed_inexact = 0
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        assert g["status"] == 0
        d = diff(r, g)
        assert d is None, f"{kind} {fname} W={width} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        relaxed_inexact += (r["comp_type"] == RELAXED and not r["is_exact"])
    if width and model.n >= 30:
        assert relaxed_inexact > 0    # merges (and their relaxed arc costs) are exercised


def test_vector_ranking_tie_break_is_the_packed_words():
    """Model::compare_states (host fringe order) == rank, then packed words from word 0 -- the order the device selects by"""
    m = ddo_amd.Mcp.from_matrix(np.array([[0, 1, -2], [1, 0, 3], [-2, 3, 0]], dtype=np.int64))

    def pack(b, depth):
        w = np.zeros(m.ws, dtype=np.uint64)
        for v, x in enumerate(b):
            w[v // 2] |= np.uint64((x & 0xFFFFFFFF) << (32 * (v & 1)))
        w[(len(b) + 1) // 2] = np.uint64(depth)
        return w

    assert m.compare(pack([1, 2, 3], 1), pack([1, 2, -4], 1)) < 0          # rank 6 < 7
    assert m.compare(pack([3, -2, 1], 1), pack([1, -2, 3], 1)) > 0          # equal rank: word 0 low half 3 > 1
    assert m.compare(pack([1, -2, 3], 1), pack([1, 2, 3], 1)) > 0           # -2 as u32 is the larger half
    assert m.compare(pack([1, 2, 3], 2), pack([1, 2, 3], 1)) > 0            # depth word last
    assert m.compare(pack([1, 2, 3], 1), pack([1, 2, 3], 1)) == 0


def _vector_golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "vector_compile_golden.json")) as f:
        return [c for c in json.load(f)["cases"] if c["width"] <= 100]


@pytest.mark.parametrize("case", _vector_golden(), ids=lambda c: c["id"])
def test_vector_models_match_golden_on_cpu(case):
    """the committed vectors (tests/golden/make_vector_golden.py) through the emulated device code; the width-5000 cases
    run on the GPU only (tests/test_gpu_vector_parity.py)"""
    from tests.parity_util import cutset_digest
    model = (ddo_amd.Max2Sat if case["kind"] == "max2sat" else ddo_amd.Mcp).read_instance(data_path(case["kind"], case["file"]))
    e = ModelEmul(model, case["width"])
    g = e.compile(case["comp_type"], case["width"], case["best_lb"], [int(x) for x in case["state"]], case["value"], case["depth"])[0]
    assert g["status"] == 0
    for k in ("is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"):
        assert g[k] == case[k], (case["id"], k, g[k], case[k])
    assert len(g["cutset"]) == case["n_cutset"] and cutset_digest(g["cutset"]) == case["cutset_digest"]
