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
