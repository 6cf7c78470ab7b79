"""CPU suite, part 3: the workgroup logic of the device kernel (ddo_amd/csrc/misp_dd_core.hpp) built as a
lock-step host emulation and checked against the oracle and the golden fixtures.  This exercises the shared
control flow (select, merge, recycled merges, local bounds, cut-set) without a GPU; it is test infrastructure
and never stands in for the HIP build -- tests/test_gpu_parity.py is the parity suite proper."""
import json
import os

import numpy as np
import pytest

from tests.conftest import data_path
from tests.dd_wire import CT_RELAXED, CT_RESTRICTED, IN_FUSED, IN_WANT_PATHS
from tests.emul_binding import Emul
from tests.parity_util import cutset_digest, diff

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "misp_compile_golden.json")


ENGINES = [1, 2]   # 1 = per-layer rebuild (misp_dd_core.hpp), 2 = in-place layers (misp_dd_inplace.hpp)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name,width,max_compiles", [
    ("johnson8-4-4", 0, 0), ("MANN_a9", 0, 0), ("brock200_2", 0, 400), ("brock200_2", 1000, 40), ("brock200_2", 1, 150),
    ("brock200_2", 3, 150), ("keller4", 7, 300), ("hamming8-4", 0, 120), ("p_hat300-1", 0, 200), ("c-fat500-1", 0, 0),
    ("brock400_1", 200, 60),
])
def test_emulation_replays_oracle_trace(oracle, name, width, max_compiles, engine):
    inst = oracle.misp(data_path("misp", name + ".clq"))
    _, recs = inst.trace_solve(width, max_compiles)
    e = Emul(inst.n, inst.rows, inst.weights, max(r["width"] for r in recs), engine=engine)
    recycled = 0
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        assert g["status"] == 0
        d = diff(r, g)
        assert d is None, f"{name} W={width} compile #{i}: {d}"
        recycled += g["recycled_merges"]
    if name == "brock200_2" and width == 0:
        assert recycled > 0  # the clean.rs:830 "recycled" merge is exercised


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("nthreads", [256, 512, 1024])
def test_emulation_is_independent_of_the_workgroup_size(oracle, nthreads, engine):
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    _, recs = inst.trace_solve(25, 60)
    e = Emul(inst.n, inst.rows, inst.weights, 25, nthreads=nthreads, engine=engine)
    for r in recs:
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        assert diff(r, g) is None


@pytest.mark.parametrize("engine", ENGINES)
def test_emulation_weighted_instance(oracle, tmp_path, engine):
    """`n` lines with negative weights (main.rs:290-297): non-unit RUB and values below the residual value."""
    rng = np.random.RandomState(11)
    n = 70
    edges = [(a, b) for a in range(n) for b in range(a + 1, n) if rng.rand() < 0.25]
    weights = rng.randint(-4, 25, size=n)
    p = tmp_path / "w.clq"
    with open(p, "w") as f:
        f.write(f"p edge {n} {len(edges)}\n")
        for i, wv in enumerate(weights):
            f.write(f"n {i + 1} {int(wv)}\n")
        for a, b in edges:
            f.write(f"e {a + 1} {b + 1}\n")
    inst = oracle.misp(str(p))
    for width in (0, 4, 16):
        _, recs = inst.trace_solve(width, 400)
        e = Emul(inst.n, inst.rows, inst.weights, max(r["width"] for r in recs), engine=engine)
        for i, r in enumerate(recs):
            g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
            assert diff(r, g) is None, (width, i, diff(r, g))


@pytest.mark.parametrize("nthreads,width,n,density,seed", [(256, 150, 150, 0.04, 5), (256, 190, 150, 0.04, 5), (1024, 500, 150, 0.04, 5),
                                                           (256, 700, 150, 0.04, 5), (256, 1500, 150, 0.04, 5), (256, 2500, 200, 0.03, 7)])
def test_emulation_layers_with_large_classes_of_ties(oracle, tmp_path, nthreads, width, n, density, seed):
    """Layers of at most 128 + nthreads candidates are selected by counting over keys staged in LDS (select_pivot of
    misp_dd_core.hpp); wider layers go through the digit rounds on the primary key.  A graph with next to no edges makes most
    candidates of a layer agree on (value, popcount): a class of ties that the width cuts in two is ranked pair by pair up to
    256 members and by digit rounds over the state words above that (the last two cases), a class kept whole needs neither."""
    rng = np.random.RandomState(seed)
    edges = [(a, b) for a in range(n) for b in range(a + 1, n) if rng.rand() < density]
    p = tmp_path / "sparse.clq"
    with open(p, "w") as f:
        f.write(f"p edge {n} {len(edges)}\n")
        for a, b in edges:
            f.write(f"e {a + 1} {b + 1}\n")
    inst = oracle.misp(str(p))
    _, recs = inst.trace_solve(width, 24 if width < 1000 else 12)
    e = Emul(inst.n, inst.rows, inst.weights, width, nthreads=nthreads, engine=1)
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        assert diff(r, g) is None, (i, diff(r, g))


@pytest.mark.parametrize("engine", ENGINES)
def test_emulation_fused_restricted_then_relaxed(oracle, engine):
    """IN_FUSED == the device half of process_one_node (parallel.rs:391-437): the relaxed DD sees the lower bound
    improved by the restricted one."""
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    root = inst.root_state()
    e = Emul(inst.n, inst.rows, inst.weights, 40, engine=engine)
    lb = -(1 << 40)
    r0, r1 = e.compile(CT_RESTRICTED, 40, lb, root, 0, 0, flags=IN_FUSED | IN_WANT_PATHS)
    o0 = inst.compile(CT_RESTRICTED, 40, lb, root, 0, 0)
    assert diff(o0, r0) is None and not r0["is_exact"]
    o1 = inst.compile(CT_RELAXED, 40, o0["best_exact_value"], root, 0, 0)
    assert diff(o1, r1) is None
    # restricted best path is a feasible independent set of that value
    chosen = [v for v, x in r0["best_path"] if x == 1]
    assert len(chosen) == r0["best_value"]
    for i, a in enumerate(chosen):
        for b in chosen[i + 1:]:
            assert (int(inst.rows[a * inst.ws + b // 64]) >> (b % 64)) & 1


def _golden_small():
    with open(GOLDEN) as f:
        return [c for c in json.load(f)["cases"] if c["width"] <= 1000]


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("case", _golden_small(), ids=lambda c: c["id"])
def test_emulation_matches_golden(oracle, case, engine):
    inst = oracle.misp(data_path("misp", case["instance"] + ".clq"))
    e = Emul(inst.n, inst.rows, inst.weights, case["width"], engine=engine)
    state = np.array([int(x) for x in case["state"]], dtype=np.uint64)
    g = e.compile(case["comp_type"], case["width"], case["best_lb"], state, case["value"], case["depth"])[0]
    for k in ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"]:
        assert g[k] == case[k], (k, g[k], case[k])
    assert len(g["cutset"]) == case["n_cutset"] and cutset_digest(g["cutset"]) == case["cutset_digest"]


@pytest.mark.parametrize("cap,nthreads", [(8, 64), (32, 64), (64, 128)])
def test_emulation_of_a_capacity_tier(oracle, monkeypatch, cap, nthreads):
    """A capacity tier (host_solver.cpp: dispatch; Engine::create_tier) = the in-place engine with node slots for layers of
    at most `cap` nodes, asked for widths far above that: every compile either ends with ST_RETRY (the DD outgrew the
    tier: the host moves it up) or equals the oracle's record."""
    monkeypatch.setenv("DDO_EMUL_TIER", "1")
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    _, recs = inst.trace_solve(100, 300)
    e = Emul(inst.n, inst.rows, inst.weights, cap, nthreads=nthreads, engine=2)
    done = retry = 0
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        if g["status"] == 78:   # ST_RETRY
            retry += 1
            continue
        assert g["status"] == 0
        assert diff(r, g) is None, f"cap {cap} compile #{i}: {diff(r, g)}"
        done += 1
    assert done > 0 and retry > 0


@pytest.mark.parametrize("packed", [1, 0], ids=["keys-packed-in-hbm", "keys-in-lds"])
@pytest.mark.parametrize("table,nthreads,expect_retry", [(1 << 20, 512, False), (128, 512, True), (128, 256, True)])
def test_emulation_of_the_dense_tier(oracle, monkeypatch, table, nthreads, expect_retry, packed):
    """The dense tier (Engine::create_tier with cap_width == max_width, 512 threads, two workgroups per CU) = the in-place
    engine at full layer capacity with a SMALLER dedup table, 8-bit select digits and at most 128 tie-break keys in LDS.  A
    layer whose table could overflow ends the compile with ST_RETRY (the host hands it to the full-width engine); every
    other compile equals the oracle's record -- including the squashed layers (the capacity tiers never squash)."""
    monkeypatch.setenv("DDO_EMUL_DENSE", str(table))
    monkeypatch.setenv("DDO_EMUL_KEYS_GLOBAL", str(packed))   # the dense kernel at W = 10 000 keeps key32 << 32 | h32 per node in HBM
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    _, recs = inst.trace_solve(100, 300)
    e = Emul(inst.n, inst.rows, inst.weights, 100, nthreads=nthreads, engine=2)
    done = retry = squashed = 0
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        if g["status"] == 78:   # ST_RETRY
            retry += 1
            continue
        assert g["status"] == 0
        assert diff(r, g) is None, f"table {table} compile #{i}: {diff(r, g)}"
        done += 1
        squashed += 0 if g["is_exact"] else 1
    assert done > 0 and squashed > 0
    assert (retry > 0) == expect_retry


@pytest.mark.parametrize("case", [c for c in _golden_small() if c["width"] >= 100][:6], ids=lambda c: c["id"])
def test_dense_tier_emulation_matches_golden(oracle, monkeypatch, case):
    """8-bit select digits and the 128-key tie-break limit of the dense tier on the golden compiles (widths up to 1000)"""
    monkeypatch.setenv("DDO_EMUL_DENSE", str(1 << 20))
    monkeypatch.setenv("DDO_EMUL_KEYS_GLOBAL", "1")
    inst = oracle.misp(data_path("misp", case["instance"] + ".clq"))
    e = Emul(inst.n, inst.rows, inst.weights, case["width"], nthreads=512, engine=2)
    state = np.array([int(x) for x in case["state"]], dtype=np.uint64)
    g = e.compile(case["comp_type"], case["width"], case["best_lb"], state, case["value"], case["depth"])[0]
    for k in ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"]:
        assert g[k] == case[k], (k, g[k], case[k])
    assert len(g["cutset"]) == case["n_cutset"] and cutset_digest(g["cutset"]) == case["cutset_digest"]


@pytest.mark.parametrize("name,width,max_compiles", [("brock200_2", 30, 80), ("brock400_1", 200, 30), ("keller4", 7, 120)])
def test_emulation_cutset_paths_as_bit_rows(oracle, name, width, max_compiles):
    """IN_PATH_BITS (dd_types.h; what ddo_mdd_compile asks the in-place engine for): the paths of the cut-set nodes leave the device
    as rows of decision bits plus the branching variables once per DD.  Expanded (tests/dd_wire.py, as ddo_mdd_drain_cutset
    does) they are the u32 rows of the plain format, node for node; and every path, replayed from the residual state, ends in
    its node's state with its node's value (Problem::transition / transition_cost, misp/main.rs:77-93)."""
    from tests.dd_wire import IN_PATH_BITS
    inst = oracle.misp(data_path("misp", name + ".clq"))
    _, recs = inst.trace_solve(width, max_compiles)
    e = Emul(inst.n, inst.rows, inst.weights, max(r["width"] for r in recs), engine=2)
    seen = 0
    for i, r in enumerate(recs):
        a = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=IN_WANT_PATHS)[0]
        b = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=IN_WANT_PATHS | IN_PATH_BITS)[0]
        assert diff(r, a) is None and diff(r, b) is None
        sa, va, ua, pa = a["cutset_raw"]
        sb, vb, ub, pb = b["cutset_raw"]
        ka = sorted(range(len(va)), key=lambda j: tuple(int(x) for x in sa[j]))
        kb = sorted(range(len(vb)), key=lambda j: tuple(int(x) for x in sb[j]))
        assert len(ka) == len(kb)
        for ja, jb in zip(ka, kb):
            assert (sa[ja] == sb[jb]).all() and va[ja] == vb[jb] and ua[ja] == ub[jb] and (pa[ja] == pb[jb]).all(), (i, ja, jb)
        for j in kb[:50]:   # replay: decisions root-first = the row reversed
            st = [int(x) for x in r["state"]] + [0] * (e.ws - len(r["state"]))
            val = r["value"]
            for x in reversed([int(x) for x in pb[j]]):
                v, d = x >> 1, x & 1
                assert not d or (st[v // 64] >> (v % 64)) & 1
                st[v // 64] &= ~(1 << (v % 64))
                if d:
                    for k in range(inst.ws):
                        st[k] &= int(inst.rows[v * inst.ws + k])
                    val += int(inst.weights[v])
            assert st[:inst.ws] == [int(x) for x in sb[j][:inst.ws]] and val == vb[j]
            seen += 1
    assert seen > 20
