"""CPU suite: the TSPTW model (examples/tsptw: BASELINE config C5) of the layer-rebuilding device engine -- variable fan-out (one
child per node that may be visited next), fuzzy relaxed states, TsptwDominance, frontier cut-set and SimpleCache -- compiled
as the lock-step host emulation and replayed against traced sequential searches of the oracle, compile by compile, in order
(cache and dominance checker are shared by all compiles of a search: the replay is stateful)."""
import pytest

import ddo_amd
from tests.conftest import data_path
from tests.dd_wire import IN_CACHE, IN_DOMINANCE, IN_FRONTIER, IN_MUST_EXPLORE, IN_WANT_PATHS
from tests.emul_binding import ModelEmul
from tests.parity_util import diff

CONFIGS = [("tsptw", False, False), ("tsptw", True, False), ("tsptw", True, True), ("tsptw+dominance", False, False), ("tsptw+dominance", True, True)]


@pytest.mark.parametrize("kind,frontier,cache", CONFIGS, ids=["lel", "frontier", "frontier+cache", "lel+dominance", "frontier+cache+dominance"])
@pytest.mark.parametrize("fname,width,max_compiles", [("N20ft301", 2, 0), ("N20ft405", 3, 0), ("N20ft308", 1, 40), ("N40ft403", 3, 80), ("N40ft207", 2, 60),
                                                      ("N60ft204", 2, 50), ("N40ft201", 0, 0), ("N20ft402", 4, 60)])
def test_tsptw_replay_of_oracle_search(oracle, fname, width, max_compiles, kind, frontier, cache):
    """width 0 = TsptwWidth(nb_vars, 1) (examples/tsptw/tests.rs:42: the root DD is then exact for these instances); small fixed
    widths force restrictions, merges (TsptwRelax::merge, relax.rs:65-191) and real branch-and-bound"""
    path = data_path("tsptw", "Langevin", fname + ".dat")
    model = ddo_amd.Tsptw.read_instance(path)
    assert model.ws == 5
    summary, recs = oracle.trace_ex(kind, path, width, max_compiles, frontier, cache)
    assert recs
    e = ModelEmul(model, max(int(r["width"]) for r in recs))
    e.keep_layers(True, 1 << 15 if cache else 0)
    if "dominance" in kind:
        e.dominance(1 << 15)
    inexact = 0
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | (IN_DOMINANCE if "dominance" in kind else 0) | (IN_FRONTIER if frontier else 0) | (IN_CACHE if cache else 0)
        if cache and r["comp_type"] == 2:
            fl |= IN_MUST_EXPLORE
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0, f"{fname} compile #{i}: status {None if g is None else g['status']}"
        d = diff(r, g)
        assert d is None, f"{fname} W={width} {kind} frontier={frontier} cache={cache} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        inexact += (r["comp_type"] == 1 and not r["is_exact"])
    if width in (2, 3) and fname in ("N20ft405", "N40ft403", "N60ft204"):
        assert inexact > 0


@pytest.mark.parametrize("kind,frontier,cache", [("tsptw", False, False), ("tsptw+dominance", True, True)], ids=["lel", "frontier+cache+dominance"])
def test_tsptw_tables_left_in_global_memory(oracle, monkeypatch, kind, frontier, cache):
    """The workgroup keeps the distance / time-window tables in LDS when they fit (EngineParams::tw_lds, n <= 190 or so on the device;
    the emulation always stages them): the same replay with the tables read where the model left them (EMUL_TW_GLOBAL; on the device the
    instances of 201 and 233 nodes of tests/test_gpu_tsptw.py take that path, DDO_HIP_TW_GLOBAL forces it)."""
    monkeypatch.setenv("EMUL_TW_GLOBAL", "1")
    test_tsptw_replay_of_oracle_search(oracle, "N40ft403", 3, 80, kind, frontier, cache)


BIG = [("AFG", "rbg067a.tw", 3, 30, 8), ("Dumas", "n80w20.001.txt", 3, 30, 8), ("AFG", "rbg125a.tw", 2, 20, 8), ("AFG", "rbg132.tw", 2, 20, 14),
       ("Dumas", "n200w20.001.txt", 2, 12, 14), ("AFG", "rbg233.tw", 2, 10, 14)]


@pytest.mark.parametrize("pools", [False, True], ids=["fixed-strides", "layer-pools"])
@pytest.mark.parametrize("kind,frontier,cache", [("tsptw", False, False), ("tsptw+dominance", True, True)], ids=["lel", "frontier+cache+dominance"])
@pytest.mark.parametrize("family,fname,width,max_compiles,words", BIG, ids=[b[1] for b in BIG])
def test_tsptw_replay_beyond_64_nodes(oracle, monkeypatch, family, fname, width, max_compiles, words, kind, frontier, cache, pools):
    """instances of the reference's resources/tsptw with 68 .. 232 nodes: the node sets take K = 2 / 4 words (the reference's Set256,
    state.rs:34-69), states 3K + 2 words, decisions 8 bits; same replay as above.  `layer-pools`: the kept layers and the arc arrays
    are carved out of per-slot pools layer by layer (run_dd: dynl; what the device does for these instances since round 4) instead of
    sitting at fixed strides."""
    if pools:
        monkeypatch.setenv("DDO_EMUL_LPOOL", "1")
    path = data_path("tsptw", family, fname)
    model = ddo_amd.Tsptw.read_instance(path)
    assert model.ws == words
    summary, recs = oracle.trace_ex(kind, path, width, max_compiles, frontier, cache)
    assert recs and len(recs[0]["state"]) == words
    e = ModelEmul(model, max(int(r["width"]) for r in recs))
    e.keep_layers(True, 1 << 15 if cache else 0)
    if "dominance" in kind:
        e.dominance(1 << 15)
    merges = 0
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | (IN_DOMINANCE if "dominance" in kind else 0) | (IN_FRONTIER if frontier else 0) | (IN_CACHE if cache else 0)
        if cache and r["comp_type"] == 2:
            fl |= IN_MUST_EXPLORE
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0, f"{fname} compile #{i}: status {None if g is None else g['status']}"
        d = diff(r, g)
        assert d is None, f"{fname} W={width} {kind} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        merges += (r["comp_type"] == 1 and not r["is_exact"])
    assert merges > 0


def test_tsptw_whole_search_of_rbg132(oracle):
    """the oracle's whole sequential search of AFG/rbg132 (131 nodes: 4-word node sets, 14-word states) in the reference's example
    configuration -- TsptwWidth(nb_vars, 1), frontier cut-set, SimpleCache, TsptwDominance: 487 compiles up to 1 834 nodes wide"""
    path = data_path("tsptw", "AFG", "rbg132.tw")
    model = ddo_amd.Tsptw.read_instance(path)
    summary, recs = oracle.trace_ex("tsptw+dominance", path, 0, 0, True, True)
    assert summary["is_exact"] and summary["best_value"] == -185240000 and len(recs) == summary["compiles"]
    e = ModelEmul(model, max(int(r["width"]) for r in recs))
    e.keep_layers(True, 1 << 22)
    e.dominance(1 << 22)
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | IN_DOMINANCE | IN_FRONTIER | IN_CACHE | (IN_MUST_EXPLORE if r["comp_type"] == 2 else 0)
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0
        d = diff(r, g)
        assert d is None, f"compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"


def test_tsptw_candidate_indices_beyond_20_bits(oracle):
    """a DD slot sized for 60 000 nodes per layer: 21 children per node make 1.26 M candidate slots, so the dedup table's entries
    (tag | candidate) carry 21-bit candidate indices instead of 20 (DDCtx::cdbits) -- the solvers size their slots like this under
    TsptwWidth (nb_vars^2 nodes); the replay is the one of the small-width test above"""
    path = data_path("tsptw", "Langevin", "N20ft405.dat")
    model = ddo_amd.Tsptw.read_instance(path)
    summary, recs = oracle.trace_ex("tsptw+dominance", path, 3, 0, True, True)
    e = ModelEmul(model, 60000)
    e.keep_layers(True, 1 << 15)
    e.dominance(1 << 15)
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | IN_DOMINANCE | IN_FRONTIER | IN_CACHE | (IN_MUST_EXPLORE if r["comp_type"] == 2 else 0)
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0
        d = diff(r, g)
        assert d is None, f"compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"


def test_tsptw_model_host_side(tmp_path):
    """instance.rs:52-109 (`(f32 * 10000.0) as usize`), model.rs:36-47 (initial state: at the depot, everything else to visit)"""
    p = tmp_path / "t.dat"
    p.write_text("# tiny\n3\n0 1.5 2\n1.5 0 1.25\n2 1.25 0\n0 100\n1 50.5\n2 60\n")
    m = ddo_amd.Tsptw.read_instance(p)
    assert m.n == 3 and m.ws == 5 and m.initial_value() == 0
    s = m.initial_state()
    assert [int(x) for x in s] == [0, 0b110, 0, 0, 0]
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Tsptw.from_arrays([[0] * 257] * 257, [0] * 257, [1] * 257)   # more than the 256 nodes of the reference's Set256
    for nodes, words in ((64, 5), (65, 8), (128, 8), (129, 14), (256, 14)):   # K = 1 / 2 / 4 words per node set, 3K + 2 per state
        big = ddo_amd.Tsptw.from_arrays([[0] * nodes] * nodes, [0] * nodes, [1] * nodes)
        assert big.ws == words
        s0 = [int(x) for x in big.initial_state()]
        k = (words - 2) // 3
        assert sum(bin(x).count("1") for x in s0[k:2 * k]) == nodes - 1 and not (s0[k] & 1) and not any(s0[:k]) and not any(s0[2 * k:])
    a, b = s.copy(), s.copy()
    b[4] = 1 << 32                                                           # deeper state ranks higher (TsptwRanking)
    assert m.compare(a, b) < 0 and m.compare(b, a) > 0 and m.compare(a, a) == 0


def test_tsptw_layer_and_arc_pools_that_overflow(oracle, monkeypatch):
    """kept layers and arcs as per-slot pools (run_dd: dynl) that a DD OUTGROWS: the compile must end with the named capacity status
    (dd_types.h: ST_ERR_LPOOL / ST_ERR_APOOL) whatever layer runs out of room -- the terminal one included, where no arc of the
    layer has been written and the backward passes must not run over them -- never with results.  The pool of the
    second run is one record short of what the DD needs in all, so it is the LAST layer that does not fit."""
    path = data_path("tsptw", "Langevin", "N20ft405.dat")
    model = ddo_amd.Tsptw.read_instance(path)
    summary, recs = oracle.trace_ex("tsptw", path, 3, 6, True, False)
    r = next(x for x in recs if x["comp_type"] == 1)
    fl = IN_WANT_PATHS | IN_FRONTIER
    monkeypatch.setenv("DDO_EMUL_LPOOL", "1")

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        e = ModelEmul(model, int(r["width"]))
        e.keep_layers(True, 0)
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        for k in env:
            monkeypatch.delenv(k)
        return g

    ok = run()
    assert ok["status"] == 0 and diff(r, ok) is None
    # the nodes all kept layers hold together: grow the pool until the compile passes; one record less then fails in the LAST layer
    lo, hi = 1, 1 << 20
    while lo < hi:
        mid = (lo + hi) // 2
        if run(DDO_EMUL_LPOOL_NODES=mid)["status"] == 0:
            hi = mid
        else:
            lo = mid + 1
    assert run(DDO_EMUL_LPOOL_NODES=lo)["status"] == 0
    for short in (lo - 1, lo // 2, 3):
        g = run(DDO_EMUL_LPOOL_NODES=short)
        assert g["status"] == -3 - 2100, (short, g["status"])
        assert not g["cutset"] and g["best_value"] is None
    lo, hi = 1, 1 << 22
    while lo < hi:
        mid = (lo + hi) // 2
        if run(DDO_EMUL_APOOL_ARCS=mid)["status"] == 0:
            hi = mid
        else:
            lo = mid + 1
    for short in (lo - 1, lo // 2, 3):
        g = run(DDO_EMUL_APOOL_ARCS=short)
        assert g["status"] in (-3 - 2200, -3 - 2100), (short, g["status"])
        assert not g["cutset"] and g["best_value"] is None
