"""Weighted MAX2SAT on the device (`-m gpu`), through the C ABI: the reference's known optima (examples/max2sat/tests.rs:
65-105, BASELINE config C3's instance family frb10-6-x with n = 60), every reported assignment re-evaluated against the
clause list by the oracle's independent evaluator.  Max2SatRanking is not a total order, so explored counts are not
comparable: optimum, proof and solution are."""
import numpy as np
import pytest

import ddo_amd
from ddo_amd import FixedWidth, NbUnassignedWidth, ParallelSolver
from tests.conftest import data_path

pytestmark = pytest.mark.gpu

SMALL = [("debug", 24), ("debug2", 13), ("pass", 54), ("tautology", 7), ("unit", 6), ("negative_wt", 4258)]
FRB10 = [("frb10-6-1", 37037), ("frb10-6-2", 38196), ("frb10-6-3", 36671), ("frb10-6-4", 38928)]


@pytest.fixture(scope="module")
def have_gpu():
    if ddo_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on an MI355X box")
    return True


def solve_and_check(oracle, path, width, threads, expected):
    model = ddo_amd.Max2Sat.read_instance(path)
    s = ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), nb_threads=threads, fringe="nodup")
    c = s.maximize()
    assert c.is_exact and c.best_value == expected
    assert s.best_lower_bound() == expected and s.best_upper_bound() == expected
    values = np.zeros(model.n, dtype=np.int64)
    sol = s.best_solution()
    assert len(sol) == model.n
    for d in sol:
        assert d.value in (1, -1)
        values[d.variable] = d.value
    import ctypes as C
    assert oracle.L.oracle_max2sat_evaluate(path.encode(), values.ctypes.data_as(C.c_void_p)) == expected
    return model, s


@pytest.mark.parametrize("name,expected", SMALL)
@pytest.mark.parametrize("width,threads", [(0, 1), (1, 4), (3, 16)])
def test_max2sat_small_known_optima(have_gpu, oracle, name, expected, width, threads):
    path = data_path("max2sat", name + ".wcnf")
    model, _ = solve_and_check(oracle, path, width, threads, expected)
    v, info = oracle.max2sat_file(path, width, 0)
    assert v == expected and model.n == info["nb_vars"] and model.initial_value() <= expected


@pytest.mark.parametrize("name,expected", FRB10)
def test_max2sat_frb10_known_optima(have_gpu, oracle, name, expected):
    """n = 60: 31 state words (the 32-word template); NbUnassignedWidth like the reference's test"""
    model, s = solve_and_check(oracle, data_path("max2sat", name + ".wcnf"), 0, 256, expected)
    assert model.n == 60 and model.ws == 31


def test_max2sat_config_c3_width_5000(have_gpu, oracle):
    """BASELINE config C3: frb10-6-1 with FixedWidth(5000)"""
    solve_and_check(oracle, data_path("max2sat", "frb10-6-1.wcnf"), 5000, 64, 37037)
