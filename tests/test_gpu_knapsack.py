"""Knapsack on the device (`-m gpu`), through the C ABI, against the CPU oracle and against known optima:
BASELINE config C1 (n = 50, W = 100) plus the reference's own known answers (examples/knapsack/tests.rs)."""
import os

import numpy as np
import pytest

import ddo_amd
from ddo_amd import FixedWidth, NbUnassignedWidth, ParallelSolver
from tests.conftest import data_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def have_gpu():
    if ddo_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on an MI355X box")
    return True


def read_kp(path):
    """examples/knapsack/main.rs:267-303 restated for the test's own checks"""
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("c")]
    n, cap = int(rows[0][0]), int(rows[0][1])
    items = [(int(r[0]), int(r[1])) for r in rows[1:1 + n]]
    return cap, [p for p, _ in items], [w for _, w in items]


def dp_optimum(cap, profit, weight):
    """textbook O(n * capacity) dynamic programme"""
    best = np.zeros(cap + 1, dtype=np.int64)
    for p, w in zip(profit, weight):
        if w <= cap:
            best[w:] = np.maximum(best[w:], best[:-w] + p) if w > 0 else best[w:] + p
    return int(best[cap])


def lcg_instance(n=50, seed=12345):
    """profit, weight in [1, 1000] from a fixed LCG, capacity = floor(sum(weight)/2)"""
    x = seed
    profit, weight = [], []
    for _ in range(n):
        x = (1103515245 * x + 12345) % (1 << 31)
        profit.append(1 + x % 1000)
        x = (1103515245 * x + 12345) % (1 << 31)
        weight.append(1 + x % 1000)
    return sum(weight) // 2, profit, weight


def check_solution(s, cap, profit, weight, value):
    sol = s.best_solution()
    taken = [d.variable for d in sol if d.value == 1]
    assert len(set(d.variable for d in sol)) == len(sol) == len(profit)
    assert sum(weight[i] for i in taken) <= cap
    assert sum(profit[i] for i in taken) == value


# optimum in the file name (examples/knapsack/tests.rs:66-127 pins the same values)
SMALL = [("f1_l-d_kp_10_269", 295), ("f2_l-d_kp_20_878", 1024), ("f3_l-d_kp_4_20", 35), ("f4_l-d_kp_4_11", 23),
         ("f6_l-d_kp_10_60", 52), ("f7_l-d_kp_7_50", 107), ("f8_l-d_kp_23_10000", 9767), ("f9_l-d_kp_5_80", 130),
         ("f10_l-d_kp_20_879", 1025)]


@pytest.mark.parametrize("name,expected", SMALL)
@pytest.mark.parametrize("width", [0, 1, 3, 100])
def test_knapsack_known_optimum_and_oracle_search(have_gpu, oracle, name, expected, width):
    if name.startswith("f8") and width in (1, 3):
        pytest.skip("tens of thousands of one-at-a-time sub-problems: minutes of launch latency, nothing new")
    path = data_path("knapsack", name)
    cap, profit, weight = read_kp(path)
    assert dp_optimum(cap, profit, weight) == expected
    model = ddo_amd.Knapsack.read_instance(path)
    assert model.n == len(profit) and model.ws == 2 and [int(x) for x in model.initial_state()] == [cap, 0]
    s = ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), nb_threads=1, fringe="nodup")
    c = s.maximize()
    assert c.is_exact and c.best_value == expected
    assert s.best_lower_bound() == expected and s.best_upper_bound() == expected
    check_solution(s, cap, profit, weight, expected)
    v, ref = oracle.knapsack_file(path, width, 1)   # ParallelSolver, one thread: the host mirrors parallel.rs
    assert v == expected
    cnt = s.counters()
    assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])


def test_readme_instance(have_gpu):
    """README / parallel.rs:902-1151: 3 items, capacity 50 => 220"""
    model = ddo_amd.Knapsack.from_items(50, [60, 100, 120], [10, 20, 30])
    s = ParallelSolver(model, FixedWidth(2), nb_threads=4, fringe="nodup")
    c = s.maximize()
    assert c.is_exact and c.best_value == 220
    check_solution(s, 50, [60, 100, 120], [10, 20, 30], 220)


@pytest.mark.parametrize("width,threads", [(100, 1), (100, 64), (10, 16), (1000, 8)])
def test_config_c1_n50(have_gpu, oracle, width, threads):
    """BASELINE config C1: n = 50 LCG instance, FixedWidth(100): proved optimum == textbook DP == oracle"""
    cap, profit, weight = lcg_instance()
    opt = dp_optimum(cap, profit, weight)
    model = ddo_amd.Knapsack.from_items(cap, profit, weight)
    s = ParallelSolver(model, FixedWidth(width), nb_threads=threads, fringe="nodup")
    c = s.maximize()
    assert c.is_exact and c.best_value == opt
    check_solution(s, cap, profit, weight, opt)
    v, ref = oracle.knapsack(profit, weight, cap, width, 1)
    assert v == opt
    if threads == 1:
        cnt = s.counters()
        assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
               (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])


@pytest.mark.parametrize("name", ["knapPI_1_100_1000_1", "knapPI_2_100_1000_1", "knapPI_3_100_1000_1"])
def test_knapsack_n100(have_gpu, oracle, name):
    path = data_path("knapsack", name)
    if not os.path.exists(path):
        pytest.skip("instance not shipped")
    cap, profit, weight = read_kp(path)
    opt = dp_optimum(cap, profit, weight)
    model = ddo_amd.Knapsack.read_instance(path)
    s = ParallelSolver(model, FixedWidth(100), nb_threads=32, fringe="nodup")
    c = s.maximize()
    assert c.is_exact and c.best_value == opt
    check_solution(s, cap, profit, weight, opt)


@pytest.mark.parametrize("name,width", [("f7_l-d_kp_7_50", 1), ("f1_l-d_kp_10_269", 3), ("f10_l-d_kp_20_879", 0)])
def test_sequential_solver_bookkeeping(have_gpu, oracle, name, width):
    """BASELINE config C1 runs the reference's SequentialSolver: same counters AND its own `explored` bookkeeping
    (sequential.rs:433-461 counts every popped node, parallel.rs:531-553 stops counting once ub <= best_lb)."""
    path = data_path("knapsack", name)
    model = ddo_amd.Knapsack.read_instance(path)
    s = ddo_amd.SequentialSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n))
    c = s.maximize()
    v, ref = oracle.knapsack_file(path, width, 0)        # nthreads = 0: the oracle's SequentialSolver
    assert c.is_exact and c.best_value == v
    cnt = s.counters()
    assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])


def test_config_c1_sequential(have_gpu, oracle):
    cap, profit, weight = lcg_instance()
    model = ddo_amd.Knapsack.from_items(cap, profit, weight)
    s = ddo_amd.SequentialSolver(model, FixedWidth(100))
    c = s.maximize()
    v, ref = oracle.knapsack(profit, weight, cap, 100, 0)
    assert c.is_exact and c.best_value == v == dp_optimum(cap, profit, weight)
    cnt = s.counters()
    assert (s.explored(), cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["explored"], ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])
