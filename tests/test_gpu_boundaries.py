"""Boundaries of the device arithmetic and of the engines.

The reference computes values as isize (64 bits, clean.rs:140-176); the device keeps them in int32 and packs the ranking key of
the in-place engine as (value - vbase) << 11 | popcount.  Two guarded limits follow:
  * sum |weights| >= 2^20: the in-place engine's 21 value bits do not suffice -> the layer-rebuilding engine compiles the DD
    (ddo_hip_engine.hip: engine_kind_ = 1);
  * sum |weights| >= 2^30: int32 values could overflow -> ddo_model_create_misp refuses loudly.
The tests sit ON both limits (one below, at it) and compare with the oracle, whose values are int64.
Also here: the in-place engine's width limit (capS = 2 W + 8 < 65 535 slots: W = 100 000 runs on the layer-rebuilding engine),
against the golden vectors of tests/golden/make_wide_golden.py.
"""
import json
import os

import numpy as np
import pytest

import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver, SubProblem
from tests.conftest import data_path
from tests.parity_util import canon_from_mdd, cutset_digest, diff, replay_records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def have_gpu():
    assert ddo_amd.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X (ddo_amd has no CPU fallback)"
    return True


def _weights_with_sum(n, total, seed):
    """n positive weights with the given sum, a few of them large (so that values really reach the high bits)"""
    rng = np.random.RandomState(seed)
    w = rng.randint(1, 1000, size=n).astype(np.int64)
    big = rng.choice(n, size=4, replace=False)
    rest = total - int(w.sum())
    assert rest > 0
    share = rest // 4
    for i in big:
        w[i] += share
    w[big[0]] += total - int(w.sum())
    assert int(w.sum()) == total and (w > 0).all()
    return w


def _write_weighted(path, n, p_edge, weights, seed):
    rng = np.random.RandomState(seed)
    edges = [(a, b) for a in range(n) for b in range(a + 1, n) if rng.rand() < p_edge]
    with open(path, "w") as f:
        f.write(f"p edge {n} {len(edges)}\n")
        for i, wv in enumerate(weights):
            f.write(f"n {i + 1} {int(wv)}\n")
        for a, b in edges:
            f.write(f"e {a + 1} {b + 1}\n")


@pytest.mark.parametrize("total", [(1 << 20) - 1, 1 << 20, (1 << 30) - 1], ids=["2^20-1", "2^20", "2^30-1"])
def test_values_at_the_engine_and_int32_limits(have_gpu, oracle, tmp_path, total):
    """One below the in-place engine's limit (in place), at it (layer rebuilding), one below the int32 guard: every compile of
    the oracle's sequential search is reproduced, values included, and the solver proves the oracle's optimum."""
    n = 48
    w = _weights_with_sum(n, total, seed=total % 9973)
    p = tmp_path / "limit.clq"
    _write_weighted(p, n, 0.25, w, seed=5)
    inst = oracle.misp(str(p))
    model = ddo_amd.Misp.read_instance(str(p))
    rows, wd = model.export()
    assert np.array_equal(wd, inst.weights) and int(np.abs(wd).sum()) == total
    for width in (3, 16):
        _, recs = inst.trace_solve(width, 40)
        assert recs
        top = max(r["best_value"] for r in recs if r["best_value"] is not None)
        assert top > total // 8          # the values do use the high bits
        for i, r, got in replay_records(model, recs):
            assert diff(r, got) is None, (total, width, i, diff(r, got))
    s = ParallelSolver(model, FixedWidth(8), nb_threads=16)
    c = s.maximize()
    assert c.is_exact and c.best_value == inst.solve(8, 0)["best_value"]


def test_weights_beyond_int32_are_refused(have_gpu):
    """sum |weights| = 2^30: the model is refused with a message, not computed with wrapped values."""
    n = 8
    ws = 1
    rows = np.full(n * ws, (1 << n) - 1, dtype=np.uint64)
    w = np.full(n, (1 << 30) // n, dtype=np.int64)
    assert int(w.sum()) == 1 << 30
    with pytest.raises(ddo_amd.DdoError, match="2\\^30"):
        ddo_amd.Misp.from_rows(n, rows, w)
    w[0] -= 1
    ok = ddo_amd.Misp.from_rows(n, rows, w)          # one below: accepted
    s = ParallelSolver(ok, FixedWidth(4), nb_threads=4)
    assert s.maximize().best_value == int(w.sum())   # complement rows all ones: no conflicts, every vertex is taken


def _wide_cases():
    with open(os.path.join(os.path.dirname(__file__), "golden", "misp_wide_golden.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _wide_cases(), ids=lambda c: c["id"])
def test_width_100000_compile_against_the_golden_vector(have_gpu, case):
    """brock400_1's root DDs at width 100 000 (beyond the in-place engine's 65 535 node slots): counters and cut-set equal
    the oracle's (fixture: the oracle needs minutes for them)."""
    model = ddo_amd.Misp.read_instance(data_path("misp", case["instance"] + ".clq"))
    mdd = ddo_amd.Mdd(model, case["width"])
    state = np.array([int(x) for x in case["state"]], dtype=np.uint64)
    sub = SubProblem(state=state, value=case["value"], path=[], depth=case["depth"])
    comp = mdd.compile(case["comp_type"], case["width"], sub, case["best_lb"])
    got = canon_from_mdd(mdd, comp, model.ws)
    for k in ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"]:
        assert got[k] == case[k], f"{case['id']}: {k} expected {case[k]} got {got[k]}"
    assert len(got["cutset"]) == case["n_cutset"]
    assert cutset_digest(got["cutset"]) == case["cutset_digest"]
