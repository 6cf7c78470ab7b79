"""Per-compile and whole-search parity of the signed-vector models (MAX2SAT = BASELINE config C3, MCP) on the device
(`-m gpu`, through the C ABI) -- the three suites MISP has:

  (1) every compile() of oracle searches replayed through ddo_mdd_compile_batch: is_exact, best_value, best_exact_value,
      nodes_expanded, arcs, layers and the cut-set multiset {state, value, ub, depth} equal;
  (2) whole searches with one sub-problem in flight: explored sub-problems and counters equal the oracle's;
  (3) committed golden fixtures (tests/golden/vector_compile_golden.json, incl. frb10-6-1 at FixedWidth(5000)).

These models' rankings (sum |benefit|) are not total orders; the reference leaves ties to its hash map's iteration order.
Oracle, host fringe and device share ONE deterministic tie-break (packed state words) and one
order-independent rule for equal-valued best arcs (an exact best path wins), which is what makes bit-exact comparison
possible; against the reference itself parity stays on optimum + proof (tests/test_gpu_max2sat.py, test_gpu_mcp.py)."""
import json
import os

import numpy as np
import pytest

import ddo_amd
from ddo_amd import FixedWidth, NbUnassignedWidth, ParallelSolver, SubProblem
from tests.conftest import data_path
from tests.parity_util import canon_from_mdd, cutset_digest, diff, replay_records

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "vector_compile_golden.json")
MODELS = {"max2sat": ddo_amd.Max2Sat, "mcp": ddo_amd.Mcp}


@pytest.fixture(scope="module")
def have_gpu():
    if ddo_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on an MI355X box")
    return True


@pytest.mark.parametrize("kind,fname,width,max_compiles", [
    ("max2sat", "pass.wcnf", 2, 0), ("max2sat", "pass.wcnf", 3, 0), ("max2sat", "debug2.wcnf", 1, 0), ("max2sat", "negative_wt.wcnf", 2, 0),
    ("max2sat", "frb10-6-1.wcnf", 8, 200), ("max2sat", "frb10-6-2.wcnf", 25, 150), ("max2sat", "frb10-6-3.wcnf", 0, 120),
    ("max2sat", "frb10-6-4.wcnf", 300, 60), ("max2sat", "frb10-6-1.wcnf", 5000, 8),
    ("mcp", "mcp_n30_p0.1_000.mcp", 3, 300), ("mcp", "mcp_n30_p0.1_001.mcp", 10, 300), ("mcp", "mcp_n30_p0.1_004.mcp", 0, 300),
    ("mcp", "mcp_n30_p0.1_007.mcp", 2, 300), ("mcp", "mcp_n30_p0.1_008.mcp", 100, 200), ("mcp", "mcp_n30_p0.1_003.mcp", 1000, 60),
    ("max2sat", "frb15-9-1.wcnf", 6, 40), ("max2sat", "frb15-9-1.wcnf", 500, 8), ("max2sat", "frb15-9-1.wcnf", 5000, 4),   # n = 135: 69 words
])
def test_replay_of_oracle_trace(have_gpu, oracle, kind, fname, width, max_compiles):
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    _, recs = oracle.vector_trace(kind, path, width, max_compiles)
    assert recs and len(recs[0]["state"]) == model.ws
    inexact = 0
    for i, r, got in replay_records(model, recs):
        d = diff(r, got)
        assert d is None, f"{kind} {fname} W={width} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        inexact += (r["comp_type"] == 1 and not r["is_exact"])
    if width and model.n >= 30:
        assert inexact > 0


@pytest.mark.parametrize("kind,fname,width", [
    ("max2sat", "pass.wcnf", 2), ("max2sat", "debug.wcnf", 1), ("max2sat", "negative_wt.wcnf", 3), ("max2sat", "unit.wcnf", 1),
    ("max2sat", "frb10-6-2.wcnf", 0), ("max2sat", "frb10-6-4.wcnf", 40),
    ("mcp", "mcp_n30_p0.1_000.mcp", 0), ("mcp", "mcp_n30_p0.1_002.mcp", 5), ("mcp", "mcp_n30_p0.1_006.mcp", 20),
    ("mcp", "mcp_n30_p0.1_009.mcp", 2),
])
def test_whole_search_with_one_subproblem_in_flight(have_gpu, oracle, kind, fname, width):
    """nb_concurrent = 1: the device-backed search pops, compiles and enqueues exactly what the oracle's one-thread
    ParallelSolver does -- same explored count, same counters, same optimum."""
    path = data_path(kind, fname)
    model = MODELS[kind].read_instance(path)
    s = ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), nb_threads=1, fringe="nodup")
    c = s.maximize()
    v, ref = (oracle.max2sat_file(path, width, 1) if kind == "max2sat" else oracle.mcp_file(path, width, 1))
    assert c.is_exact and c.best_value == v == ref["best_value"]
    cnt = s.counters()
    assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])


def _golden_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["id"])
def test_golden_compile(have_gpu, case):
    model = MODELS[case["kind"]].read_instance(data_path(case["kind"], case["file"]))
    mdd = ddo_amd.Mdd(model, case["width"])
    sub = SubProblem(state=np.array([int(x) for x in case["state"]], dtype=np.uint64), value=case["value"], path=[], depth=case["depth"])
    comp = mdd.compile(case["comp_type"], case["width"], sub, case["best_lb"])
    got = canon_from_mdd(mdd, comp, model.ws)
    for k in ("is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"):
        assert got[k] == case[k], (case["id"], k, got[k], case[k])
    assert len(got["cutset"]) == case["n_cutset"] and cutset_digest(got["cutset"]) == case["cutset_digest"]
    # every cut-set node carries the decisions that lead to it: replaying them from the residual state is not
    # possible without the model's transition on the host, but their count must be the node's depth
    for n in got["cutset_nodes"]:
        assert len(n.path) == n.depth - case["depth"]


def test_frb15_stress_instance_bounds(have_gpu):
    """BASELINE config C3's stress instance frb15-9-1 (n = 135, 2648 clauses, optimum 341783 -- examples/max2sat/tests.rs:107-110,
    ignored there as too slow): 69-word states on the 72-word template.  Under a time budget the search stops with
    incumbent <= optimum <= best open bound, and the incumbent is a real assignment of that weight."""
    import ctypes as C
    from tests.oracle_binding import Oracle
    import os
    from tests.conftest import ROOT
    path = data_path("max2sat", "frb15-9-1.wcnf")
    model = ddo_amd.Max2Sat.read_instance(path)
    assert model.n == 135 and model.ws == 69
    s = ParallelSolver(model, FixedWidth(1000), ddo_amd.TimeBudget(8.0), nb_threads=64, fringe="nodup")
    c = s.maximize()
    assert c.best_value is not None and s.best_lower_bound() <= 341783 <= s.best_upper_bound()
    values = np.zeros(model.n, dtype=np.int64)
    for d in s.best_solution():
        values[d.variable] = d.value
    o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    assert o.L.oracle_max2sat_evaluate(path.encode(), values.ctypes.data_as(C.c_void_p)) == c.best_value
