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


@pytest.mark.parametrize("name,width,max_compiles", [
    ("johnson8-4-4", 0, 0), ("MANN_a9", 0, 0), ("brock200_2", 0, 400), ("brock200_2", 1000, 40), ("brock200_2", 1, 150),
    ("brock200_2", 3, 150), ("keller4", 7, 300), ("hamming8-4", 0, 120), ("p_hat300-1", 0, 200), ("c-fat500-1", 0, 0),
    ("brock400_1", 200, 60),
])
def test_emulation_replays_oracle_trace(oracle, name, width, max_compiles):
    inst = oracle.misp(data_path("misp", name + ".clq"))
    _, recs = inst.trace_solve(width, max_compiles)
    e = Emul(inst.n, inst.rows, inst.weights, max(r["width"] for r in recs))
    recycled = 0
    for i, r in enumerate(recs):
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        assert g["status"] == 0
        d = diff(r, g)
        assert d is None, f"{name} W={width} compile #{i}: {d}"
        recycled += g["recycled_merges"]
    if name == "brock200_2" and width == 0:
        assert recycled > 0  # the clean.rs:830 "recycled" merge is exercised


@pytest.mark.parametrize("nthreads", [256, 512, 1024])
def test_emulation_is_independent_of_the_workgroup_size(oracle, nthreads):
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    _, recs = inst.trace_solve(25, 60)
    e = Emul(inst.n, inst.rows, inst.weights, 25, nthreads=nthreads)
    for r in recs:
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"])[0]
        assert diff(r, g) is None


def test_emulation_fused_restricted_then_relaxed(oracle):
    """IN_FUSED == the device half of process_one_node (parallel.rs:391-437): the relaxed DD sees the lower bound
    improved by the restricted one."""
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    root = inst.root_state()
    e = Emul(inst.n, inst.rows, inst.weights, 40)
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


@pytest.mark.parametrize("case", _golden_small(), ids=lambda c: c["id"])
def test_emulation_matches_golden(oracle, case):
    inst = oracle.misp(data_path("misp", case["instance"] + ".clq"))
    e = Emul(inst.n, inst.rows, inst.weights, case["width"])
    state = np.array([int(x) for x in case["state"]], dtype=np.uint64)
    g = e.compile(case["comp_type"], case["width"], case["best_lb"], state, case["value"], case["depth"])[0]
    for k in ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"]:
        assert g[k] == case[k], (k, g[k], case[k])
    assert len(g["cutset"]) == case["n_cutset"] and cutset_digest(g["cutset"]) == case["cutset_digest"]
