"""The drop-in plug end to end (`-m gpu`): the ORACLE's restatement of the reference's host -- SequentialSolver / ParallelSolver with
their worker threads, the mutex-shared NoDupFringe, MaxUB, the width heuristics (sequential.rs:202-527, parallel.rs:287-641) -- runs
with the device engine as its `DecisionDiagram` (tests/shim/hip_mdd_shim.cpp: `HipMdd`, every method one call of include/ddo_hip.h;
the compiled C++ twin of hip_mdd/src/lib.rs, which no toolchain here can build).  north_star's configuration: "host and
branch-and-bound fringe stay [the reference's], calling HIP through a thin extern "C" FFI".

* sequential: the search through HipMdd is THE SAME SEARCH as through the oracle's own Mdd -- explored sub-problems, compiles, nodes,
  arcs and layers are equal, because every compile returns the same bits;
* parallel: T worker threads call compile() concurrently on their own mdds (parallel.rs:576-602); the optimum is proved, the
  solution is an independent set of that weight, and the compiles shared device launches (the combining layer);
* the same with Pooled decision diagrams (hip_mdd::install_pooled)."""
import pytest

import ddo_amd
from tests.conftest import data_path
from tests.parity_util import is_independent_set
from tests.shim_binding import shim_misp_solve

pytestmark = pytest.mark.gpu


def _feasible(name, sol, value):
    model = ddo_amd.Misp.read_instance(data_path("misp", name + ".clq"))
    rows, weights = model.export()
    taken = [v for v, x in sol if x == 1]
    return is_independent_set(rows, model.ws, taken) and int(sum(weights[v] for v in taken)) == value


@pytest.mark.parametrize("name,width,expected", [("johnson8-4-4", 0, 14), ("MANN_a9", 0, 16), ("hamming6-4", 0, 4), ("keller4", 0, 11),
                                                 ("brock200_2", 100, 12), ("p_hat300-1", 50, 8), ("brock200_2", 3000, 12)])
def test_the_reference_sequential_solver_over_hipmdd_is_the_same_search(oracle, name, width, expected):
    path = data_path("misp", name + ".clq")
    ref = oracle.misp(path).solve(width, 0)
    got = shim_misp_solve(path, width, 0)
    assert got["is_exact"] and got["best_value"] == expected == ref["best_value"]
    for k in ("explored", "compiles", "nodes_expanded", "arcs", "layers", "best_lb", "best_ub"):
        assert got[k] == ref[k], (name, width, k, got[k], ref[k])
    assert _feasible(name, got["solution"], expected)


@pytest.mark.parametrize("name,width,threads,expected", [("keller4", 0, 16, 11), ("brock200_2", 100, 64, 12), ("brock200_4", 200, 128, 17),
                                                         ("p_hat300-1", 50, 32, 8), ("brock200_2", 3000, 64, 12)])
def test_the_reference_parallel_solver_over_hipmdd(name, width, threads, expected):
    path = data_path("misp", name + ".clq")
    got = shim_misp_solve(path, width, threads)
    assert got["is_exact"] and got["best_value"] == expected and got["best_lb"] == got["best_ub"] == expected
    assert _feasible(name, got["solution"], expected)
    assert got["requests"] >= got["compiles"] > 0
    # concurrent compiles of the worker threads share launches (one launch per compile would be requests == launches)
    if got["compiles"] > 20 * threads:
        assert got["requests"] > 2 * got["launches"], got
    print(f"{name} W={width} T={threads}: {got['compiles']} compiles in {got['launches']} launches, explored {got['explored']}, {got['wall_s']:.2f} s")


@pytest.mark.parametrize("name,width", [("johnson8-4-4", 5), ("MANN_a9", 20), ("keller4", 200)])
def test_the_reference_sequential_solver_over_pooled_hipmdd(oracle, name, width):
    path = data_path("misp", name + ".clq")
    ref = oracle.misp(path).solve(width, 0, pooled=True)
    got = shim_misp_solve(path, width, 0, pooled=True)
    assert got["is_exact"] and got["best_value"] == ref["best_value"]
    for k in ("explored", "compiles", "nodes_expanded", "arcs", "layers"):
        assert got[k] == ref[k], (name, width, k, got[k], ref[k])


@pytest.mark.parametrize("name,width,pooled", [("brock200_2", 300, False), ("keller4", 40, False), ("keller4", 40, True)])
def test_the_bulk_drain_hands_over_what_the_callbacks_deliver(name, width, pooled):
    """ddo_mdd_drain_cutset_rows (the shims' one call per relaxed compile, made OUTSIDE the solver's lock) against ddo_mdd_drain_cutset:
    the same nodes -- state, value, ub, depth, and the path once the residual's own path is put in front; `ub_above`
    leaves out exactly the nodes whose bound does not exceed it; one drain per compile, whichever of the two."""
    import numpy as np
    from ddo_amd import CompilationType, Decision, SubProblem
    model = ddo_amd.Misp.read_instance(data_path("misp", name + ".clq"))
    make = (lambda: ddo_amd.Pooled(model, width)) if pooled else (lambda: ddo_amd.Mdd(model, width))
    a, b, c = make(), make(), make()
    root = model.root()
    v = model.n - 1
    head = [Decision(v, 0)]   # the residual: the root after decision "vertex n - 1 stays out"
    words = [int(w) for w in root.state]
    words[v >> 6] &= ~(1 << (v & 63))
    state = np.array(words, dtype=np.uint64)
    sub = SubProblem(state=state, value=0, path=head, ub=1 << 40, depth=len(head))
    for m in (a, b, c):
        comp = m.compile(CompilationType.Relaxed, width, sub, -(1 << 40))
        assert not comp.is_exact
    n = a.cutset_count()
    assert n > 10 and b.cutset_count() == n
    ref = a.drain_cutset()
    assert len(ref) == n and a.cutset_count() == 0 and a.drain_cutset() == []
    rows = b.drain_cutset_rows(residual_path=head)
    assert b.cutset_count() == 0 and b.drain_cutset() == []

    def key(s):
        # (a pooled node keeps ONE of its equal-valued paths, and which one differs between two compiles: paths are compared on the default DD)
        return (tuple(int(w) for w in s.state), int(s.value), int(s.ub), int(s.depth), () if pooled else tuple((d.variable, d.value) for d in s.path))
    assert all(s.path[:len(head)] == head or [(d.variable, d.value) for d in s.path[:len(head)]] == [(d.variable, d.value) for d in head] for s in rows)
    # (two compiles of one input list their cut-set rows in different orders: the rows' positions come from device atomics)
    assert sorted(key(s) for s in rows) == sorted(key(s) for s in ref)
    ubs = sorted(int(s.ub) for s in ref)
    cut = ubs[len(ubs) // 2]
    some = c.drain_cutset_rows(ub_above=cut, residual_path=head)
    assert sorted(key(s) for s in some) == sorted(key(s) for s in ref if int(s.ub) > cut) and len(some) < n


@pytest.mark.parametrize("name,width,pooled", [("johnson8-4-4", 4, False), ("MANN_a9", 3, False), ("hamming6-4", 6, False), ("brock200_2", 100, False),
                                               ("MANN_a9", 0, True), ("hamming6-4", 5, True), ("keller4", 0, True)])
def test_the_reference_sequential_solver_over_hipmdd_and_hipcache_is_the_same_search(oracle, name, width, pooled):
    """`SequentialSolver<BitSet, HipMdd, HipCache>`: the mdds take the device-side SimpleCache (DDO_MDD_CACHING, ddo_compile_input.cache) and the
    solver's own Cache calls -- must_explore at the pop, clear -- reach the same table through the ABI's host views (`impl Cache for
    HipCache`, hip_mdd/src/lib.rs; here its C++ twin).  The search is the oracle's cached search: SeqCachingSolverLel (solver/mod.rs:45) and,
    over Pooled decision diagrams, SeqCachingSolverPooled (:47) -- same explored sub-problems, same optimum, proof."""
    path = data_path("misp", name + ".clq")
    ref, _ = oracle.trace_ex("misp+pooled" if pooled else "misp", path, width, 0, False, True)
    plain, _ = oracle.trace_ex("misp+pooled" if pooled else "misp", path, width, 0, False, False)
    # (a table large enough never to refuse an entry: a pooled search caches every exact node it expands -- keller4 fills 2^18 entries,
    # explores 6 666 sub-problems instead of 5 625 and is still right: a refused threshold is less pruning)
    got = shim_misp_solve(path, width, 0, pooled=pooled, cache_entries=1 << 22)
    assert got["is_exact"] and got["best_value"] == ref["best_value"] == plain["best_value"]
    assert got["explored"] == ref["explored"], (name, width, pooled, got["explored"], ref["explored"], plain["explored"])
    assert _feasible(name, got["solution"], ref["best_value"])


@pytest.mark.parametrize("name,width,threads,pooled,expected", [("keller4", 0, 16, False, 11), ("brock200_2", 100, 32, False, 12), ("MANN_a9", 0, 16, True, 16)])
def test_the_reference_parallel_solver_over_hipmdd_and_hipcache(name, width, threads, pooled, expected):
    path = data_path("misp", name + ".clq")
    got = shim_misp_solve(path, width, threads, pooled=pooled, cache_entries=1 << 18)
    assert got["is_exact"] and got["best_value"] == expected and got["best_lb"] == got["best_ub"] == expected
    assert _feasible(name, got["solution"], expected)
