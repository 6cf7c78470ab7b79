"""CPU suite: the POOLED variant of the in-place device engine (misp_dd_inplace.hpp: run_dd2<WS, DEEP, POOLED = 1>; the reference's
long-arc decision diagram, mdd/pooled.rs:117-823) compiled as the lock-step host emulation and compared with the oracle's
Pooled<S> -- compile by compile over traced SeqNoCachingSolverPooled searches (solver/mod.rs:43) and on the 48 single compiles of
tests/golden/misp_pooled_golden.json: is_exact, best value, best exact value, nodes / arcs / layers, and the FRONTIER cut-set as a
multiset of (state, value, ub, depth) -- depth being the layer at which the node was expanded.  tests/test_gpu_pooled.py runs the
same comparisons on the device through the C ABI."""
import json
import os

import numpy as np
import pytest

from tests.conftest import data_path
from tests.dd_wire import IN_WANT_PATHS
from tests.emul_binding import Emul
from tests.parity_util import cutset_digest, diff

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "misp_pooled_golden.json")
POOL = 14000   # node slots of a pooled engine slot on the device (Engine::init: pool_nodes)


def _cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["id"])
def test_pooled_emulation_matches_the_golden_compiles(oracle, case):
    inst = oracle.misp(data_path("misp", case["instance"] + ".clq"))
    e = Emul(inst.n, inst.rows, inst.weights, 2000, engine=2)
    e.pooled(True)
    state = np.array([int(x) for x in case["state"]], dtype=np.uint64)
    g = e.compile(case["comp_type"], case["width"], case["best_lb"], state, case["value"], case["depth"], flags=IN_WANT_PATHS)[0]
    assert g["status"] == 0
    for k in ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"]:
        assert g[k] == case[k], (k, g[k], case[k])
    assert len(g["cutset"]) == case["n_cutset"] and cutset_digest(g["cutset"]) == case["cutset_digest"]


@pytest.mark.parametrize("name,width,max_compiles", [
    ("johnson8-4-4", 5, 200), ("brock200_2", 5, 150), ("brock200_2", 50, 60), ("MANN_a9", 5, 200), ("hamming6-4", 5, 120), ("keller4", 7, 200),
    ("p_hat300-1", 20, 60), ("brock200_4", 0, 60), ("hamming8-4", 0, 40), ("c-fat500-1", 0, 0),
])
def test_pooled_emulation_replays_an_oracle_search(oracle, name, width, max_compiles):
    """width 0 = NbUnassignedWidth, which counts a sub-problem's PATH -- shorter than its depth in a pooled search (one decision per
    expanded ancestor, pooled.rs:316-334): the trace carries the widths the oracle's solver asked for"""
    inst = oracle.misp(data_path("misp", name + ".clq"))
    _, recs = inst.trace_solve(width, max_compiles, pooled=True)
    assert recs
    e = Emul(inst.n, inst.rows, inst.weights, POOL, engine=2)
    e.pooled(True)
    squashed = cut = 0
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=IN_WANT_PATHS)[0]
        assert g["status"] == 0, (i, g["status"])
        d = diff(r, g)
        assert d is None, f"{name} W={width} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        squashed += not r["is_exact"]
        cut += len(r["cutset"])
        # a cut-set node's path, replayed from the residual state over the layers whose variable impacts it, ends in the node
        sb, vb, ub, pb = g["cutset_raw"]
        for j in range(min(len(vb), 20)):
            st = [int(x) for x in r["state"]] + [0] * (e.ws - len(r["state"]))
            val = r["value"]
            depth = int(g["cs_depth"][j])
            for x in [int(x) for x in pb[j]][::-1][:depth]:   # (rows are node first over the DD's whole stride: reversed = layer 0 first)
                v, dec = x >> 1, x & 1
                if not (st[v // 64] >> (v % 64)) & 1:
                    assert not dec
                    continue
                st[v // 64] &= ~(1 << (v % 64))
                if dec:
                    for k in range(inst.ws):
                        st[k] &= int(inst.rows[v * inst.ws + k])
                    val += int(inst.weights[v])
            assert st[:inst.ws] == [int(x) for x in sb[j][:inst.ws]] and val == vb[j], (i, j)
    if width and name != "p_hat300-1":
        assert squashed > 0 and cut > 0


def test_a_pool_that_outgrows_its_slots_is_a_capacity_error(oracle):
    """a Pooled DD's pool is not bounded by the width (only its layers are): a slot holds as many nodes as the engine gives it, and a
    pool beyond that ends the compile with a capacity status -- never with results"""
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    e = Emul(inst.n, inst.rows, inst.weights, 300, engine=2)   # 608 node slots
    e.pooled(True)
    g = e.compile(1, 200, -(1 << 40), inst.root_state(), 0, 0, flags=IN_WANT_PATHS)[0]
    assert g["status"] <= -100 and not g["cutset"] and g["best_value"] is None


def _weighted_instance(tmp_path, n=70, seed=11, p_edge=0.25):
    rng = np.random.RandomState(seed)
    edges = [(a, b) for a in range(n) for b in range(a + 1, n) if rng.rand() < p_edge]
    weights = rng.randint(-4, 25, size=n)
    p = tmp_path / "w.clq"
    with open(p, "w") as f:
        f.write(f"p edge {n} {len(edges)}\n")
        for i, wv in enumerate(weights):
            f.write(f"n {i + 1} {int(wv)}\n")
        for a, b in edges:
            f.write(f"e {a + 1} {b + 1}\n")
    return str(p)


def test_pooled_emulation_on_a_weighted_instance(oracle, tmp_path):
    """`n` lines with negative weights (main.rs:290-297): the rough upper bound of a node is the sum of the weights of its vertices
    (not its popcount), values fall below the residual value, and a cut-set node's value is rebuilt from the weights along its path"""
    inst = oracle.misp(_weighted_instance(tmp_path))
    for width in (0, 4, 16):
        _, recs = inst.trace_solve(width, 300, pooled=True)
        e = Emul(inst.n, inst.rows, inst.weights, 4000, engine=2)
        e.pooled(True)
        for i, r in enumerate(recs):
            g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=IN_WANT_PATHS)[0]
            assert g["status"] == 0 and diff(r, g) is None, (width, i, g["status"], diff(r, g))


@pytest.mark.parametrize("nthreads", [64, 256, 512, 1024])
def test_pooled_emulation_is_independent_of_the_workgroup_size(oracle, nthreads):
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    _, recs = inst.trace_solve(25, 40, pooled=True)
    e = Emul(inst.n, inst.rows, inst.weights, 4000, nthreads=nthreads, engine=2)
    e.pooled(True)
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=IN_WANT_PATHS)[0]
        assert g["status"] == 0 and diff(r, g) is None, (i, diff(r, g))


def test_pooled_emulation_exact_compile_and_a_lower_bound_that_prunes_everything(oracle):
    """CompilationType::Exact: no width, no cut-set, the optimum of the sub-problem; and a compile whose best_lb lies above every
    rough upper bound: no node is expanded beyond the root, no best value (Completion { best_value: None })"""
    inst = oracle.misp(data_path("misp", "hamming6-4.clq"))
    e = Emul(inst.n, inst.rows, inst.weights, 6000, engine=2)
    e.pooled(True)
    root = inst.root_state()
    for comp_type, width, lb in ((0, 6000, -(1 << 40)), (1, 5, 1000), (2, 5, 1000)):
        ref = inst.compile(comp_type, width if comp_type else (1 << 62), lb, root, 0, 0, pooled=True)
        g = e.compile(comp_type, width, lb, root, 0, 0, flags=IN_WANT_PATHS)[0]
        assert g["status"] == 0 and diff(ref, g) is None, (comp_type, diff(ref, g))
    assert inst.compile(0, 1 << 62, -(1 << 40), root, 0, 0, pooled=True)["best_value"] == 4


@pytest.mark.parametrize("name,width,max_compiles", [("johnson8-4-4", 4, 300), ("johnson8-4-4", 0, 200), ("keller4", 7, 200), ("brock200_2", 10, 100),
                                                     ("MANN_a9", 3, 300), ("hamming6-4", 5, 150), ("p_hat300-1", 20, 60), ("brock200_4", 0, 40)])
def test_pooled_emulation_replays_a_search_behind_a_simple_cache(oracle, name, width, max_compiles):
    """SeqCachingSolverPooled (solver/mod.rs:47): Pooled decision diagrams behind a SimpleCache -- _filter_with_cache on the impacted nodes
    of every layer but the first (pooled.rs:635, 662-680), _compute_thresholds + _maybe_update_cache over the long arcs (:467-535).  The
    replay is stateful like the default DD's (tests/test_emulation_cache.py): a threshold written by compile k decides what compile
    k + 1 prunes, so every record only matches when the device-side table holds what the reference's holds after each compile."""
    from tests.dd_wire import IN_CACHE, IN_MUST_EXPLORE
    path = data_path("misp", name + ".clq")
    inst = oracle.misp(path)
    plain, _ = oracle.trace_ex("misp+pooled", path, width, max_compiles, False, False)
    summary, recs = oracle.trace_ex("misp+pooled", path, width, max_compiles, False, True)
    assert recs
    e = Emul(inst.n, inst.rows, inst.weights, POOL, engine=2)
    e.pooled(True)
    e.pooled_cache(1 << 16)
    hits = 0
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | IN_CACHE | (IN_MUST_EXPLORE if r["comp_type"] == 2 else 0)
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0, (i, None if g is None else g["status"])
        d = diff(r, g)
        assert d is None, f"{name} W={width} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        hits += g["cache_hits"]
    assert e.cache_used() > 0
    print(name, width, "compiles", len(recs), "cache hits", hits, "explored", summary["explored"], "vs", plain["explored"])
