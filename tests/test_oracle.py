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
    assert p.stdout.count("\nok ") + p.stdout.startswith("ok ") >= 41
    # every DD case runs twice: on the default DD (clean.rs's tests) and on Pooled (the same-named tests of pooled.rs:1024-2250)
    assert p.stdout.count("[Pooled, pooled.rs]") == 14 and p.stdout.count("[Mdd, clean.rs]") == 14
    assert "ok pooled_long_arcs_on_misp" in p.stdout


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


# hamming8-2 is left out: the pooled relaxation of its root is weaker (134 against the default DD's exact 128 -- sound, but the
# search that follows runs for minutes where the default DD proves the optimum at the root: MISP branches on the variable the
# fewest states contain, so a pooled layer stays below the width for a long time, nothing is merged, the pool grows -- and meets
# the small widths NbUnassignedWidth leaves for the last layers all at once); brock200_2 is in the next test
@pytest.mark.parametrize("name,expected", sorted((k, v) for k, v in MISP_KATS.items() if k not in ("brock200_2", "hamming8-2")))
@pytest.mark.parametrize("nthreads", [0, 4], ids=["sequential", "4-threads"])
def test_misp_known_optimum_over_pooled_dds(oracle, name, expected, nthreads):
    """the optima of examples/misp/tests.rs through Seq / ParNoCachingSolverPooled (solver/mod.rs:34, :43): Pooled DDs
    (mdd/pooled.rs), whose layers hold only the nodes the branching variable impacts (misp/main.rs:145-147)"""
    inst = oracle.misp(data_path("misp", name + ".clq"))
    r = inst.solve(0, nthreads, pooled=True)
    assert r["is_exact"] and r["best_value"] == expected
    assert r["best_lb"] == expected == r["best_ub"]
    chosen = [v for v, x in r["solution"] if x == 1]
    assert len(chosen) == expected
    for i, a in enumerate(chosen):
        for b in chosen[i + 1:]:
            assert (int(inst.rows[a * inst.ws + b // 64]) >> (b % 64)) & 1


def test_pooled_search_expands_fewer_nodes_than_the_default_dd(oracle):
    """the point of Pooled on MISP: nodes the variable does not impact are not copied layer after layer"""
    for name, expected in (("c-fat500-1", 14), ("c-fat200-1", 12), ("johnson8-4-4", 14)):
        inst = oracle.misp(data_path("misp", name + ".clq"))
        a, b = inst.solve(0, 0), inst.solve(0, 0, pooled=True)
        assert a["best_value"] == b["best_value"] == expected
        assert 5 * b["nodes_expanded"] < a["nodes_expanded"], (name, a["nodes_expanded"], b["nodes_expanded"])
    # ... not a law: on brock200_2 the pooled search explores more sub-problems (weaker merges) and as many nodes
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    assert inst.solve(0, 0, pooled=True)["best_value"] == 12


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


# ---- TSPTW: oracle (with SimpleCache, frontier cut-set and SimpleDominanceChecker) pinned on examples/tsptw/tests.rs ----
TSPTW_KAT = [
    ("N20ft301", 661.6), ("N20ft302", 703.0), ("N20ft303", 746.4), ("N20ft304", 817.0), ("N20ft305", 724.7), ("N20ft306", 729.5),
    ("N20ft307", 691.8), ("N20ft308", 788.2), ("N20ft309", 751.8), ("N20ft310", 693.8), ("N20ft401", 660.9), ("N20ft402", 701.0),
    ("N20ft403", 746.4), ("N20ft404", 817.0), ("N20ft405", 724.7), ("N20ft406", 728.5), ("N20ft407", 691.8), ("N20ft408", 786.1),
    ("N20ft409", 749.8), ("N20ft410", 693.8), ("N40ft201", 1109.3), ("N40ft202", 1017.4), ("N40ft203", 903.1), ("N40ft204", 897.4),
    ("N40ft205", 983.6), ("N40ft206", 1081.9), ("N40ft207", 884.9), ("N40ft208", 1051.6), ("N40ft209", 1027.5), ("N40ft210", 1035.3),
    ("N40ft401", 1105.2), ("N40ft402", 1016.4), ("N40ft403", 903.1), ("N40ft404", 897.4), ("N40ft405", 982.6), ("N40ft406", 1081.9),
    ("N40ft407", 872.2), ("N40ft408", 1043.5), ("N40ft409", 1025.5), ("N40ft410", 1034.3), ("N60ft201", 1375.4), ("N60ft202", 1186.4),
    ("N60ft203", 1194.2), ("N60ft204", 1283.6), ("N60ft205", 1215.5), ("N60ft206", 1238.8), ("N60ft207", 1305.3), ("N60ft208", 1172.6),
    ("N60ft209", 1243.8), ("N60ft210", 1273.2), ("N60ft301", 1375.4), ("N60ft302", 1184.4), ("N60ft303", 1194.2), ("N60ft304", 1283.6),
    ("N60ft305", 1214.5), ("N60ft306", 1237.8), ("N60ft307", 1298.4), ("N60ft308", 1168.8), ("N60ft309", 1242.8), ("N60ft310", 1273.2),
    ("N60ft401", 1375.4), ("N60ft402", 1183.4), ("N60ft403", 1194.2), ("N60ft404", 1283.6), ("N60ft405", 1212.5), ("N60ft406", 1236.8),
    ("N60ft407", 1296.4), ("N60ft408", 1150.0), ("N60ft409", 1241.8), ("N60ft410", 1273.2),
]


@pytest.mark.parametrize("name,expected", TSPTW_KAT)
def test_tsptw_langevin(oracle, name, expected):
    """tests.rs:81-499, every non-ignored Langevin case (`solve_langevin`: width factor 1, one thread):
    tour length = -value / 10000, compared in f32 like the reference does"""
    path = data_path("tsptw", "Langevin", name + ".dat")
    v, info = oracle.tsptw_file(path, 1, 1)
    n = int(name[1:3])
    assert info["nb_nodes"] == n and info["has_value"]
    assert np.float32(-v) / np.float32(10000.0) == np.float32(expected)
    assert info["tour_length"] == -v and info["tour"][-1] == 0 and sorted(info["tour"]) == list(range(n))


def test_tsptw_config_c5_instance(oracle):
    """tests.rs:201-203, the BASELINE config C5 instance family (n = 40): 1109.30"""
    v, info = oracle.tsptw_file(data_path("tsptw", "Langevin", "N40ft201.dat"), 1, 4)
    assert np.float32(-v) / np.float32(10000.0) == np.float32(1109.30)
    assert info["tour_length"] == -v


def _pooled_cases():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "misp_pooled_golden.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _pooled_cases(), ids=lambda c: c["id"])
def test_pooled_compiles_against_the_golden_vectors(oracle, case):
    """single compiles as POOLED DDs (oracle Pooled<S>, pooled.rs:117-823) against tests/golden/misp_pooled_golden.json
    (made by tests/golden/make_pooled_golden.py): the fixture a device variant of the pooled DD has to hit."""
    import numpy as np
    from tests.parity_util import cutset_digest
    inst = oracle.misp(data_path("misp", case["instance"] + ".clq"))
    state = np.array([int(x) for x in case["state"]], dtype=np.uint64)
    r = inst.compile(case["comp_type"], case["width"], case["best_lb"], state, case["value"], case["depth"], pooled=True)
    for k in ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"]:
        assert r[k] == case[k], (case["id"], k, r[k], case[k])
    assert len(r["cutset"]) == case["n_cutset"] and cutset_digest(r["cutset"]) == case["cutset_digest"]
    assert [list(p) for p in r["best_path"]] == case["best_path"]
    # what any DD of the sub-problem must satisfy: the best path is an independent set worth best_value - value
    if r["best_value"] is not None:
        chosen = [v for v, x in r["best_path"] if x == 1]
        assert int(sum(inst.weights[v] for v in chosen)) == r["best_value"] - case["value"] or not r["is_exact"]
