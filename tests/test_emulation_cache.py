"""CPU suite: frontier cut-set (clean.rs:586-606), thresholds (clean.rs:478-545) and SimpleCache (cache/simple.rs:36-73) of the
layer-rebuilding device engine (ddo_amd/csrc/misp_dd_core.hpp + dd_thresholds.hpp), compiled as the lock-step host emulation.

The oracle runs a traced SEQUENTIAL search with the reference's solver aliases (solver/mod.rs): DefaultMDDLEL or DefaultMDDFC,
EmptyCache or SimpleCache.  Every compile of that search is replayed, in order, through the emulated device code with its
own cache table: a search with a cache is stateful -- a threshold written by compile k decides what compile k + 1 prunes --
so every record (values, node / arc / layer counters, cut-set multiset with per-node depths) only matches when the
device-side cache holds exactly what the reference's holds after each compile."""
import pytest

import ddo_amd
from tests.conftest import data_path
from tests.dd_wire import IN_CACHE, IN_DOMINANCE, IN_FRONTIER, IN_MUST_EXPLORE, IN_WANT_PATHS
from tests.emul_binding import ModelEmul
from tests.parity_util import diff

MODELS = {"misp": ddo_amd.Misp, "knapsack": ddo_amd.Knapsack, "max2sat": ddo_amd.Max2Sat, "mcp": ddo_amd.Mcp}
CASES = [("misp", "johnson8-4-4.clq", 4, 600), ("misp", "brock200_2.clq", 10, 60), ("misp", "keller4.clq", 7, 80), ("misp", "MANN_a9.clq", 3, 0),
         ("knapsack", "f1_l-d_kp_10_269", 3, 0), ("knapsack", "f8_l-d_kp_23_10000", 5, 60), ("max2sat", "pass.wcnf", 2, 0),
         ("max2sat", "frb10-6-1.wcnf", 8, 30), ("mcp", "mcp_n30_p0.1_000.mcp", 3, 50), ("mcp", "mcp_n30_p0.1_005.mcp", 12, 40)]


@pytest.mark.parametrize("frontier,cache", [(True, False), (False, True), (True, True)], ids=["frontier", "lel+cache", "frontier+cache"])
@pytest.mark.parametrize("kind,fname,width,max_compiles", CASES)
def test_replay_of_a_search_with_frontier_cutset_and_cache(oracle, kind, fname, width, max_compiles, frontier, cache):
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    summary, recs = oracle.trace_ex(kind, path, width, max_compiles, frontier, cache)
    assert recs
    e = ModelEmul(model, max(int(r["width"]) for r in recs))
    e.keep_layers(True, 1 << 15 if cache else 0)
    hits = several_depths = 0
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | (IN_FRONTIER if frontier else 0) | (IN_CACHE if cache else 0)
        if cache and r["comp_type"] == 2:
            fl |= IN_MUST_EXPLORE      # sequential.rs:341: the oracle explored this node, so must the device-side cache say
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0, f"{kind} {fname} compile #{i}: status {None if g is None else g['status']}"
        d = diff(r, g)
        assert d is None, f"{kind} {fname} W={width} frontier={frontier} cache={cache} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        hits += g["cache_hits"]
        several_depths += len({c[3] for c in r["cutset"]}) > 1
    if cache:
        assert e.cache_used() > 0
    if frontier and kind == "misp" and max_compiles != 0:
        assert several_depths > 0      # a frontier cut-set has nodes of several layers


def test_cache_prunes_and_changes_the_search(oracle):
    """the cache is not a no-op: with it the sequential frontier-cut-set search of johnson8-4-4 at width 4 explores fewer sub-problems, and the
    emulated device code removes nodes by _filter_with_cache (clean.rs:710-726)"""
    path = data_path("misp", "johnson8-4-4.clq")
    plain, _ = oracle.trace_ex("misp", path, 4, 0, True, False)
    cached, recs = oracle.trace_ex("misp", path, 4, 0, True, True)
    assert cached["best_value"] == plain["best_value"] == 14 and cached["explored"] < plain["explored"]
    model = ddo_amd.Misp.read_instance(path)
    e = ModelEmul(model, 4)
    e.keep_layers(True, 1 << 15)
    hits = 0
    for r in recs:
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"],
                      flags=IN_WANT_PATHS | IN_FRONTIER | IN_CACHE | (IN_MUST_EXPLORE if r["comp_type"] == 2 else 0))[0]
        assert g["status"] == 0 and diff(r, g) is None
        hits += g["cache_hits"]
    assert hits > 0


@pytest.mark.parametrize("frontier,cache", [(False, False), (True, False), (False, True), (True, True)],
                         ids=["lel", "frontier", "lel+cache", "frontier+cache"])
@pytest.mark.parametrize("fname,width,max_compiles", [("f1_l-d_kp_10_269", 3, 0), ("f8_l-d_kp_23_10000", 3, 300), ("f8_l-d_kp_23_10000", 20, 200),
                                                      ("knapPI_1_100_1000_1", 3, 0), ("f10_l-d_kp_20_879", 5, 0), ("f7_l-d_kp_7_50", 2, 0)])
def test_replay_of_a_knapsack_search_with_dominance(oracle, fname, width, max_compiles, frontier, cache):
    """SimpleDominanceChecker(KPDominance) (dominance/simple.rs:37-117, examples/knapsack/main.rs:198-218, 325) -- with frontier
    cut-set and cache this is the reference's own knapsack configuration (SeqCachingSolverFc).  The checker is shared by all
    compiles of a search, so the replay is stateful like the cache's: _filter_with_dominance (clean.rs:689-708) of compile k
    sees the non-dominated pairs compiles 0..k-1 left."""
    path = data_path("knapsack", fname)
    model = ddo_amd.Knapsack.read_instance(path)
    plain, _ = oracle.trace_ex("knapsack", path, width, max_compiles, frontier, cache)
    summary, recs = oracle.trace_ex("knapsack+dominance", path, width, max_compiles, frontier, cache)
    if max_compiles == 0:
        assert summary["best_value"] == plain["best_value"] and summary["nodes_expanded"] <= plain["nodes_expanded"]
    e = ModelEmul(model, max(int(r["width"]) for r in recs))
    e.keep_layers(True, 1 << 15 if cache else 0)
    e.dominance(2048)
    removed = 0
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | IN_DOMINANCE | (IN_FRONTIER if frontier else 0) | (IN_CACHE if cache else 0)
        if cache and r["comp_type"] == 2:
            fl |= IN_MUST_EXPLORE
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0
        d = diff(r, g)
        assert d is None, f"{fname} W={width} frontier={frontier} cache={cache} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        removed += g["cache_hits"]
    if fname.startswith("f8") and width == 20:
        assert removed > 0      # nodes did leave layers (dominated, or pruned by the cache)
