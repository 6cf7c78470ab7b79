"""CPU suite, part 1: the oracle against the reference's known answers (oracle pinning).

Cites: ddo/src/implementation/mdd/clean.rs:1153-2398 (engine KATs, via oracle/kat_main.cpp),
ddo/examples/misp/tests.rs:71-161, ddo/examples/knapsack/tests.rs:66-127, README.md:246-292."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import data_path


def test_engine_fringe_solver_kats(oracle_build):
    """oracle/kat_main.cpp restates the reference's unit tests (Dummy / LocB fixtures, NoDupFringe,
    widths, knapsack solver KATs); every case must print ok."""
    p = subprocess.run([os.path.join(oracle_build, "kat")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ALL OK" in p.stdout
    assert p.stdout.count("\nok ") + p.stdout.startswith("ok ") >= 26


# the fast subset of examples/misp/tests.rs (NbUnassignedWidth, NoDupFringe, default solver)
MISP_KATS = {"brock200_2": 12, "c-fat200-1": 12, "c-fat200-2": 24, "c-fat200-5": 58, "c-fat500-1": 14, "c-fat500-2": 26,
             "hamming6-2": 32, "hamming6-4": 4, "hamming8-2": 128, "johnson8-2-4": 4, "johnson8-4-4": 14,
             "MANN_a9": 16, "p_hat300-1": 8}


@pytest.mark.parametrize("name,expected", sorted(MISP_KATS.items()))
def test_misp_known_optimum(oracle, name, expected):
    inst = oracle.misp(data_path("misp", name + ".clq"))
    r = inst.solve(0, 0)  # sequential
    assert r["is_exact"] and r["best_value"] == expected
    assert r["best_lb"] == expected == r["best_ub"]
    chosen = [v for v, x in r["solution"] if x == 1]
    assert len(chosen) == expected  # unit weights
    for i, a in enumerate(chosen):  # feasibility check of examples/misp/main.rs:381-388
        for b in chosen[i + 1:]:
            assert (int(inst.rows[a * inst.ws + b // 64]) >> (b % 64)) & 1


@pytest.mark.slow
@pytest.mark.parametrize("name,expected", [("keller4", 11), ("brock200_3", 15)])
def test_misp_known_optimum_parallel(oracle, name, expected):
    r = oracle.misp(data_path("misp", name + ".clq")).solve(0, 4)  # ParallelSolver, 4 threads
    assert r["is_exact"] and r["best_value"] == expected


def test_misp_fixed_width_gives_same_optimum(oracle):
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    for w in (1, 7, 1000):
        assert inst.solve(w, 0)["best_value"] == 12


KNAPSACK_KATS = {"f9_l-d_kp_5_80": 130, "f7_l-d_kp_7_50": 107, "f3_l-d_kp_4_20": 35, "f4_l-d_kp_4_11": 23,
                 "f10_l-d_kp_20_879": 1025, "f1_l-d_kp_10_269": 295, "f6_l-d_kp_10_60": 52, "f8_l-d_kp_23_10000": 9767,
                 "f2_l-d_kp_20_878": 1024, "knapPI_1_100_1000_1": 9147, "knapPI_2_100_1000_1": 1514,
                 "knapPI_3_100_1000_1": 2397}


@pytest.mark.parametrize("name,expected", sorted(KNAPSACK_KATS.items()))
def test_knapsack_known_optimum(oracle, name, expected):
    v, info = oracle.knapsack_file(data_path("knapsack", name), 0, 0)
    assert v == expected and info["is_exact"]


def test_knapsack_readme_example(oracle):
    """README.md:246-292 / parallel.rs:902-940: capacity 50, profits 60/100/120, weights 10/20/30 -> 220, x=(0,1,1)."""
    for width in (0, 1, 100):
        for threads in (0, 2):
            v, info = oracle.knapsack([60, 100, 120], [10, 20, 30], 50, width, threads)
            assert v == 220 and info["solution"] == [0, 1, 1]


def test_knapsack_n50_width100_against_dp(oracle):
    """BASELINE config C1: n = 50 (seeded LCG), FixedWidth(100), SequentialSolver; the optimum
    must equal a textbook O(n*C) dynamic program."""
    seed = 12345
    profit, weight = [], []
    for _ in range(50):
        seed = (seed * 1103515245 + 12345) % (1 << 31)
        profit.append(1 + seed % 1000)
        seed = (seed * 1103515245 + 12345) % (1 << 31)
        weight.append(1 + seed % 1000)
    cap = sum(weight) // 2
    best = np.zeros(cap + 1, dtype=np.int64)
    for p, w in zip(profit, weight):
        best[w:] = np.maximum(best[w:], best[:-w] + p)
    v, info = oracle.knapsack(profit, weight, cap, 100, 0)
    assert v == int(best[cap]) and info["is_exact"]
    taken = [i for i, x in enumerate(info["solution"]) if x == 1]
    assert sum(weight[i] for i in taken) <= cap and sum(profit[i] for i in taken) == v


# ---- MAX2SAT (the next model family): oracle pinned on examples/max2sat/tests.rs:65-105 and data.rs:119-126 ---------
MAX2SAT_KAT = [("debug", 24), ("debug2", 13), ("pass", 54), ("tautology", 7), ("unit", 6), ("negative_wt", 4258)]


@pytest.mark.parametrize("name,expected", MAX2SAT_KAT)
@pytest.mark.parametrize("width,threads", [(0, 0), (1, 0), (3, 2)])
def test_max2sat_small_known_optima(oracle, name, expected, width, threads):
    v, info = oracle.max2sat_file(data_path("max2sat", name + ".wcnf"), width, threads)
    assert v == expected and info["is_exact"]
    assert info["best_lb"] == expected and info["best_ub"] == expected
    assert info["solution_weight"] == expected      # the decisions really satisfy clauses worth the optimum


def test_max2sat_reader(oracle):
    """data.rs:119-126: debug2.wcnf has 3 variables and 4 clauses (a unit clause written `w x x 0` included)"""
    _, info = oracle.max2sat_file(data_path("max2sat", "debug2.wcnf"))
    assert (info["nb_vars"], info["nb_clauses"]) == (3, 4)
    _, info = oracle.max2sat_file(data_path("max2sat", "frb10-6-1.wcnf"), 2, 0, 0.2)
    assert info["nb_vars"] == 60


@pytest.mark.parametrize("name,expected", [("frb10-6-1", 37037), ("frb10-6-2", 38196), ("frb10-6-3", 36671),
                                           ("frb10-6-4", 38928)])
def test_max2sat_frb10_known_optima(oracle, name, expected):
    """examples/max2sat/tests.rs:89-104 (n = 60, BASELINE config C3's parity instance family)"""
    v, info = oracle.max2sat_file(data_path("max2sat", name + ".wcnf"), 0, 16)
    assert v == expected and info["is_exact"] and info["solution_weight"] == expected


# ---- MCP (maximum cut): oracle pinned on examples/mcp/tests.rs:64-103 ----------------------------------------------
MCP_KAT = [("000", 13), ("001", 18), ("002", 15), ("003", 19), ("004", 16), ("005", 19), ("006", 12), ("007", 18),
           ("008", 20), ("009", 22)]


@pytest.mark.parametrize("idx,expected", MCP_KAT)
def test_mcp_known_optima(oracle, idx, expected):
    v, info = oracle.mcp_file(data_path("mcp", f"mcp_n30_p0.1_{idx}.mcp"), 0, 8)
    assert info["nb_vertices"] == 30
    assert v == expected and info["is_exact"]
    assert info["cut_weight"] == expected and info["solution"][0] == 1     # the first vertex is fixed on side S


def test_mcp_sequential_and_fixed_width(oracle):
    for width, threads in ((0, 0), (5, 0), (50, 2)):
        v, info = oracle.mcp_file(data_path("mcp", "mcp_n30_p0.1_006.mcp"), width, threads)
        assert v == 12 and info["cut_weight"] == 12
