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
