"""The drop-in boundary under the reference's own threading contract (`-m gpu`).  ddo's ParallelSolver owns one DecisionDiagram
per worker thread and calls compile() from all of them CONCURRENTLY (parallel.rs:576-602; SURVEY.md section 8 b1, "Threading").
Here T host threads (tools/b1_driver.cpp: the reference's process_one_node, parallel.rs:391-437, over include/ddo_hip.h only)
loop plain `ddo_mdd_compile` on sub-problems of brock400_1's root cut-set at width 10 000 (BASELINE config C4's instance and width):

* every compile's digest -- is_exact, best value, best exact value, nodes / arcs / layers, the cut-set as an order-independent
  checksum of (state, value, ub, depth) -- equals the CPU oracle's for that sub-problem (tests/golden/b1_brock400_golden.json,
  generator beside it), whichever launch it went out with and whoever its neighbours were;
* the concurrent compiles SHARE launches (Engine::compile_combined): the mean number of decision diagrams per device launch is at
  least 0.8 x min(T, node slots) -- one launch per compile would run the 256 CUs one workgroup at a time."""
import json
import os

import numpy as np
import pytest

import ddo_amd
from ddo_amd.boundary import cutset_hash, run_b1
from tests.conftest import data_path

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DENSE_SLOTS = 512   # two 512-thread workgroups per CU x 256 CUs (Engine::init: the dense tier)


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "b1_brock400_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def brock400():
    return ddo_amd.Misp.read_instance(data_path("misp", "brock400_1.clq"))


def _subs(gold):
    subs = gold["subproblems"]
    return (np.array([s["state"] for s in subs], dtype=np.uint64), [s["value"] for s in subs], [s["depth"] for s in subs])


@pytest.mark.parametrize("threads,items_per_thread", [(64, 16), (512, 12), (2048, 4)])
def test_concurrent_compiles_share_launches_and_equal_the_oracle(gold, brock400, threads, items_per_thread):
    states, values, depths = _subs(gold)
    tot, r0, r1 = run_b1(brock400, states, values, depths, gold["width"], threads, threads * items_per_thread, gold["best_lb"])
    mean = tot["requests"] / max(1, tot["launches"])
    print(f"T={threads}: {tot['compiles']} compiles ({tot['requests']} with the ones that ran again) in {tot['launches']} launches = {mean:.1f} decision "
          f"diagrams per launch; {tot['nodes_expanded'] / tot['seconds']:.3e} nodes/s through ddo_mdd_compile ({tot['seconds']:.2f} s, kernels "
          f"{tot['kernel_ms'] / 1e3:.2f} s; {tot['cutset_nodes']} cut-set nodes, {tot['path_decisions']} path decisions drained; thread-seconds in "
          f"compile {tot['compile_s']:.1f}, in drain {tot['drain_s']:.1f})")
    assert tot["errors"] == 0, tot
    assert tot["mismatches"] == 0, tot   # every later compile of a sub-problem equals its first one ...
    for i, s in enumerate(gold["subproblems"]):   # ... and the first one equals the oracle's
        assert r0[i] == s["restricted"], (i, r0[i], s["restricted"])
        assert r1[i] == s["relaxed"], (i, r1[i], s["relaxed"])
    assert tot["requests"] >= tot["compiles"] and tot["launches"] > 0   # (a compile that found the shared output arena full runs again)
    slots = 256 if os.environ.get("DDO_HIP_NO_AUTO_DENSE") else DENSE_SLOTS   # (diagnosis switch: the full-width engine's slots, one per CU)
    assert mean >= 0.8 * min(threads, slots), (mean, tot)


def test_the_cutset_checksum_is_the_drivers(gold, brock400):
    """the Python restatement of the driver's checksum (ddo_amd/boundary.py: cutset_hash, used by the golden generator) against the
    driver's own on one relaxed compile drained through the Python binding"""
    states, values, depths = _subs(gold)
    k = next(i for i, s in enumerate(gold["subproblems"]) if s["relaxed"] and s["relaxed"]["n_cutset"] > 0)
    states, values, depths = states[k:k + 1], values[k:k + 1], depths[k:k + 1]
    tot, r0, r1 = run_b1(brock400, states, values, depths, gold["width"], 1, 1, gold["best_lb"])
    mdd = ddo_amd.Mdd(brock400, gold["width"])
    sub = ddo_amd.SubProblem(state=states[0], value=values[0], path=[], depth=depths[0])
    lb = max(gold["best_lb"], r0[0]["best_exact_value"]) if r0[0]["has_best_exact"] else gold["best_lb"]
    comp = mdd.compile(ddo_amd.CompilationType.Relaxed, gold["width"], sub, lb)
    cut = [(tuple(int(x) for x in n.state), int(n.value), int(n.ub), int(n.depth)) for n in mdd.drain_cutset()]
    assert not comp.is_exact and len(cut) == r1[0]["n_cutset"] and cutset_hash(cut) == r1[0]["cutset_hash"]
