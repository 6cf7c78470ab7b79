"""Frontier cut-set, thresholds and SimpleCache on the device (`-m gpu`, through the C ABI): SURVEY.md section 8 rows a14, a16, f1.

  (1) DefaultMDDFC: every compile of oracle searches with a frontier cut-set replayed through ddo_mdd_compile_batch
      (stateless: EmptyCache) -- values, counters, cut-set multiset with per-node depths and path lengths;
  (2) a SimpleCache in device memory behind ddo_mdd_compile: the compiles of a cached oracle search replayed IN ORDER
      against one ddo_cache -- they only match when the table holds after every compile what the reference's holds;
  (3) whole searches: the device-backed SequentialSolver with DefaultMDDLEL / DefaultMDDFC and Empty / Simple cache
      reproduces the oracle's explored count and counters; DefaultCachingSolver with many sub-problems in flight proves
      the known optima;
  (4) Cache::get_threshold / update_threshold through the host views of the table."""
import numpy as np
import pytest

import ddo_amd
from ddo_amd import FRONTIER, LAST_EXACT_LAYER, DefaultCachingSolver, FixedWidth, NbUnassignedWidth, SequentialSolver, SubProblem
from tests.conftest import data_path
from tests.parity_util import canon_from_mdd, diff

pytestmark = pytest.mark.gpu
MODELS = {"misp": ddo_amd.Misp, "knapsack": ddo_amd.Knapsack, "max2sat": ddo_amd.Max2Sat, "mcp": ddo_amd.Mcp}
CASES = [("misp", "johnson8-4-4.clq", 4, 400), ("misp", "brock200_2.clq", 10, 120), ("misp", "brock200_2.clq", 100, 60), ("misp", "keller4.clq", 7, 150),
         ("knapsack", "f1_l-d_kp_10_269", 3, 0), ("knapsack", "f8_l-d_kp_23_10000", 5, 100), ("max2sat", "pass.wcnf", 2, 0),
         ("max2sat", "frb10-6-1.wcnf", 8, 60), ("max2sat", "frb10-6-2.wcnf", 200, 30), ("mcp", "mcp_n30_p0.1_000.mcp", 3, 80), ("mcp", "mcp_n30_p0.1_005.mcp", 12, 60)]


@pytest.fixture(scope="module")
def have_gpu():
    if ddo_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on an MI355X box")
    return True


def _sub(r):
    return SubProblem(state=np.array(r["state"], dtype=np.uint64), value=r["value"], path=[], depth=r["depth"])


@pytest.mark.parametrize("kind,fname,width,max_compiles", CASES)
def test_frontier_cutset_replay(have_gpu, oracle, kind, fname, width, max_compiles):
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    _, recs = oracle.trace_ex(kind, path, width, max_compiles, True, False)
    maxw = max(int(r["width"]) for r in recs)
    mdds = [ddo_amd.DefaultMDDFC(model, maxw) for _ in range(min(32, len(recs)))]
    several = 0
    for base in range(0, len(recs), len(mdds)):
        chunk = recs[base:base + len(mdds)]
        ms = mdds[:len(chunk)]
        comps = ddo_amd.Mdd.compile_batch(ms, [r["comp_type"] for r in chunk], [r["width"] for r in chunk], [_sub(r) for r in chunk],
                                          [r["best_lb"] for r in chunk])
        for j, r in enumerate(chunk):
            got = canon_from_mdd(ms[j], comps[j], model.ws)
            d = diff(r, got)
            assert d is None, f"{kind} {fname} W={width} compile #{base + j} type={r['comp_type']}: {d}"
            for n in got["cutset_nodes"]:
                assert len(n.path) == n.depth - r["depth"]
            several += len({c[3] for c in r["cutset"]}) > 1
    if kind == "misp" and max_compiles:
        assert several > 0


@pytest.mark.parametrize("frontier", [False, True], ids=["lel", "frontier"])
@pytest.mark.parametrize("kind,fname,width,max_compiles", CASES)
def test_cached_search_replayed_in_order(have_gpu, oracle, kind, fname, width, max_compiles, frontier):
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    _, recs = oracle.trace_ex(kind, path, width, max_compiles, frontier, True)
    maxw = max(int(r["width"]) for r in recs)
    mdd = ddo_amd.Mdd(model, maxw, cutset_type=FRONTIER if frontier else LAST_EXACT_LAYER, caching=True)
    cache = ddo_amd.SimpleCache(model, 1 << 16)
    for i, r in enumerate(recs):
        comp = mdd.compile(r["comp_type"], r["width"], _sub(r), r["best_lb"], cache=cache)
        d = diff(r, canon_from_mdd(mdd, comp, model.ws))
        assert d is None, f"{kind} {fname} W={width} frontier={frontier} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
    st = cache.stats()
    assert st["used"] > 0 and st["dropped"] == 0


# sequential searches of a few thousand sub-problems at most (one launch per compile: about a millisecond each)
SEQ_CASES = [("misp", "johnson8-4-4.clq", 4, True, False), ("misp", "johnson8-4-4.clq", 4, False, True), ("misp", "johnson8-4-4.clq", 4, True, True),
             ("misp", "MANN_a9.clq", 3, True, True), ("misp", "hamming6-4.clq", 6, True, True),
             ("knapsack", "f8_l-d_kp_23_10000", 5, False, True), ("knapsack", "f8_l-d_kp_23_10000", 5, True, True),
             ("max2sat", "pass.wcnf", 2, True, False), ("max2sat", "pass.wcnf", 2, True, True), ("max2sat", "frb10-6-3.wcnf", 0, False, True),
             ("mcp", "mcp_n30_p0.1_002.mcp", 4, True, False), ("mcp", "mcp_n30_p0.1_002.mcp", 4, False, True), ("mcp", "mcp_n30_p0.1_002.mcp", 4, True, True)]


@pytest.mark.parametrize("kind,fname,width,frontier,cache", SEQ_CASES)
def test_sequential_solver_matches_the_oracle(have_gpu, oracle, kind, fname, width, frontier, cache):
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    ref, _ = oracle.trace_ex(kind, path, width, 0, frontier, cache)
    s = SequentialSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), cutset_type=FRONTIER if frontier else LAST_EXACT_LAYER,
                         cache_entries=(1 << 18) if cache else 0)
    c = s.maximize()
    assert c.is_exact and c.best_value == ref["best_value"]
    cnt = s.counters()
    assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])


@pytest.mark.parametrize("kind,fname,width,frontier", [("misp", "johnson8-4-4.clq", 4, True), ("misp", "MANN_a9.clq", 3, True),
                                                       ("knapsack", "f8_l-d_kp_23_10000", 5, False), ("mcp", "mcp_n30_p0.1_002.mcp", 4, True),
                                                       ("max2sat", "pass.wcnf", 2, True)])
def test_cached_search_with_an_overflowing_output_arena(have_gpu, oracle, monkeypatch, kind, fname, width, frontier):
    """A 1 KB output arena: nearly every relaxed compile finds it too small and is compiled AGAIN with the arena enlarged
    (Engine::run_solo_growing).  A compile that fails on the arena must leave nothing in the cache -- its thresholds would
    prune the second run (and a node already marked explored would be dropped at the second pop): the search must be the
    oracle's, sub-problem by sub-problem."""
    monkeypatch.setenv("DDO_HIP_ARENA_KB", "1")
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    ref, _ = oracle.trace_ex(kind, path, width, 0, frontier, True)
    # (a width no other test uses: the engine of this (model, width, features) must be created under the small arena)
    s = SequentialSolver(model, FixedWidth(width), cutset_type=FRONTIER if frontier else LAST_EXACT_LAYER, cache_entries=1 << 18)
    c = s.maximize()
    assert c.is_exact and c.best_value == ref["best_value"]
    cnt = s.counters()
    assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])
    p = DefaultCachingSolver(model, FixedWidth(width), nb_threads=16, cache_entries=1 << 18)   # many in flight + the pop's explored mark
    c = p.maximize()
    assert c.is_exact and c.best_value == ref["best_value"] and p.best_upper_bound() == ref["best_value"]


@pytest.mark.parametrize("kind,fname,expected,width,threads", [
    ("misp", "brock200_2.clq", 12, 100, 64), ("misp", "johnson8-4-4.clq", 14, 6, 16), ("knapsack", "knapPI_1_100_1000_1", 9147, 30, 16),
    ("knapsack", "f8_l-d_kp_23_10000", 9767, 5, 32), ("max2sat", "pass.wcnf", 54, 2, 8), ("mcp", "mcp_n30_p0.1_003.mcp", None, 20, 32),
])
def test_default_caching_solver_proves_the_optimum(have_gpu, oracle, kind, fname, expected, width, threads):
    """DefaultCachingSolver = ParallelSolver<DefaultMDDFC, SimpleCache> (solver/mod.rs) with many sub-problems in flight"""
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    if expected is None:
        expected = oracle.mcp_file(path, 0, 1)[0]
    plain = ddo_amd.ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), nb_threads=threads)
    assert plain.maximize().best_value == expected
    s = DefaultCachingSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), nb_threads=threads, cache_entries=1 << 20)
    c = s.maximize()
    assert c.is_exact and c.best_value == expected and s.best_upper_bound() == expected
    assert len(s.best_solution()) == model.n or kind == "misp"


def test_cache_thresholds_through_the_host_views(have_gpu):
    """Cache::update_threshold keeps the larger Threshold under (value, explored) (simple.rs:64-68); states of different depths
    are different keys; Cache::clear empties the table"""
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock200_2.clq"))
    cache = ddo_amd.SimpleCache(model, 1024)
    a, b = model.initial_state(), model.initial_state()
    b[0] ^= np.uint64(5)
    assert cache.get_threshold(a, 3) is None
    cache.update_threshold(a, 3, 7, False)
    assert cache.get_threshold(a, 3) == (7, False) and cache.get_threshold(a, 4) is None and cache.get_threshold(b, 3) is None
    cache.update_threshold(a, 3, 7, True)
    assert cache.get_threshold(a, 3) == (7, True)
    cache.update_threshold(a, 3, 6, True)       # smaller: ignored
    cache.update_threshold(a, 3, 7, False)      # same value, not explored: ignored
    assert cache.get_threshold(a, 3) == (7, True)
    cache.update_threshold(a, 3, 9, False)
    assert cache.get_threshold(a, 3) == (9, False)
    cache.update_threshold(b, 3, -4, True)
    assert cache.get_threshold(b, 3) == (-4, True) and cache.stats()["used"] == 2
    cache.update_threshold(b, 9, (1 << 63) - 1, True)       # isize::MAX ("large theta for dangling nodes")
    assert cache.get_threshold(b, 9) == ((1 << 63) - 1, True)
    cache.clear()
    assert cache.get_threshold(a, 3) is None and cache.stats()["used"] == 0


# ---- SimpleDominanceChecker: the reference's knapsack configuration ----------------------------------
@pytest.mark.parametrize("frontier,cache", [(False, False), (True, True)], ids=["lel", "frontier+cache"])
@pytest.mark.parametrize("fname,width,max_compiles", [("f8_l-d_kp_23_10000", 3, 300), ("f8_l-d_kp_23_10000", 20, 200), ("knapPI_1_100_1000_1", 3, 0)])
def test_knapsack_dominance_replayed_in_order(have_gpu, oracle, fname, width, max_compiles, frontier, cache):
    path = data_path("knapsack", fname)
    model = ddo_amd.Knapsack.read_instance(path)
    _, recs = oracle.trace_ex("knapsack+dominance", path, width, max_compiles, frontier, cache)
    mdd = ddo_amd.Mdd(model, max(int(r["width"]) for r in recs), cutset_type=FRONTIER if frontier else LAST_EXACT_LAYER, caching=True)
    ch = ddo_amd.SimpleCache(model, 1 << 16) if cache else None
    dom = ddo_amd.SimpleDominanceChecker(model, 4096)
    for i, r in enumerate(recs):
        comp = mdd.compile(r["comp_type"], r["width"], _sub(r), r["best_lb"], cache=ch, dominance=dom)
        d = diff(r, canon_from_mdd(mdd, comp, model.ws))
        assert d is None, f"{fname} W={width} frontier={frontier} cache={cache} compile #{i}: {d}"


@pytest.mark.parametrize("fname,width", [("f8_l-d_kp_23_10000", 3), ("f8_l-d_kp_23_10000", 20), ("knapPI_1_100_1000_1", 3), ("f1_l-d_kp_10_269", 2),
                                          ("knapPI_2_100_1000_1", 10)])
def test_seq_caching_solver_fc_with_dominance(have_gpu, oracle, fname, width):
    """examples/knapsack/main.rs:320-337: SeqCachingSolverFc (frontier cut-set + SimpleCache) with SimpleDominanceChecker(KPDominance)
    -- explored sub-problems and counters equal the oracle's, and fewer nodes are expanded than without dominance"""
    path = data_path("knapsack", fname)
    model = ddo_amd.Knapsack.read_instance(path)
    ref, _ = oracle.trace_ex("knapsack+dominance", path, width, 0, True, True)
    s = SequentialSolver(model, FixedWidth(width), cutset_type=FRONTIER, cache_entries=1 << 18, dominance_entries=1 << 12)
    c = s.maximize()
    assert c.is_exact and c.best_value == ref["best_value"]
    cnt = s.counters()
    assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])
    plain = SequentialSolver(model, FixedWidth(width), cutset_type=FRONTIER, cache_entries=1 << 18)
    assert plain.maximize().best_value == c.best_value and plain.counters()["nodes_expanded"] >= cnt["nodes_expanded"]
    sol = s.best_solution()
    assert sol is not None and len(sol) == model.n


def test_dominance_is_for_knapsack_models(have_gpu):
    misp = ddo_amd.Misp.read_instance(data_path("misp", "johnson8-2-4.clq"))
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.SimpleDominanceChecker(misp, 64)
