"""The command-line front ends (ddo_amd/cli.py) against the report of the reference's example binaries
(examples/misp/main.rs:391-397 and siblings)."""
import io
import os
from contextlib import redirect_stdout

import pytest

from ddo_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")


def run(argv):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = cli.main(argv)
    lines = buf.getvalue().strip().splitlines()
    return rc, {ln.split(":", 1)[0]: ln.split(":", 1)[1].strip() for ln in lines}, lines


def test_wcnf_reader_matches_reference_counts():
    # data.rs:118-125: debug2.wcnf has 3 variables and 4 distinct clauses
    n, weights = cli._read_wcnf(os.path.join(DATA, "max2sat", "debug2.wcnf"))
    assert n == 3 and len(weights) == 4
    n, weights = cli._read_wcnf(os.path.join(DATA, "max2sat", "frb10-6-1.wcnf"))
    assert n == 60 and all(a <= b for a, b in weights)


def test_argument_surface():
    with pytest.raises(SystemExit):
        cli.main(["max2sat"])            # -f is mandatory (max2sat/main.rs:22-23)
    with pytest.raises(SystemExit):
        cli.main(["misp"])               # positional instance file (misp/main.rs:225-226)


@pytest.mark.gpu
def test_misp_report():
    rc, kv, lines = run(["misp", os.path.join(DATA, "misp", "brock200_2.clq"), "-w", "1000", "-t", "64"])
    assert rc == 0
    assert [ln.split(":")[0] for ln in lines] == ["Duration", "Objective", "Upper Bnd", "Lower Bnd", "Gap", "Aborted", "Solution"]
    assert kv["Objective"] == "12" and kv["Upper Bnd"] == "12" and kv["Lower Bnd"] == "12"
    assert kv["Gap"] == "0.000" and kv["Aborted"] == "false"
    assert len(eval(kv["Solution"])) == 12


@pytest.mark.gpu
def test_misp_report_lazy_fringe_and_default_width():
    rc, kv, _ = run(["misp", os.path.join(DATA, "misp", "keller4.clq"), "--fringe", "lazy"])
    assert rc == 0 and kv["Objective"] == "11" and kv["Aborted"] == "false"


@pytest.mark.gpu
def test_knapsack_report():
    rc, kv, _ = run(["knapsack", os.path.join(DATA, "knapsack", "f1_l-d_kp_10_269"), "-w", "100"])
    assert rc == 0 and kv["Objective"] == "295" and kv["Aborted"] == "false"
    taken = eval(kv["Solution"])
    assert len(taken) == 10 and set(taken) <= {0, 1}
    rc, kv, _ = run(["knapsack", os.path.join(DATA, "knapsack", "f3_l-d_kp_4_20")])   # FixedWidth(2) by default
    assert rc == 0 and kv["Objective"] == "35"


@pytest.mark.gpu
def test_max2sat_report_cost_is_the_falsified_weight():
    path = os.path.join(DATA, "max2sat", "frb10-6-1.wcnf")
    rc, kv, lines = run(["max2sat", "-f", path, "-w", "100", "--concurrent", "512"])
    assert rc == 0 and kv["Aborted"] == "false"
    assert [ln.split(":")[0] for ln in lines][-2:] == ["Cost", "Solution"]
    _n, weights = cli._read_wcnf(path)
    # the objective is the satisfied weight: total - falsified (max2sat/main.rs:89-118)
    assert int(kv["Objective"]) + int(kv["Cost"]) == sum(weights.values())
    lits = eval(kv["Solution"])
    assert sorted(abs(x) for x in lits) == list(range(1, 61))


@pytest.mark.gpu
def test_mcp_report():
    rc, kv, _ = run(["mcp", "-f", os.path.join(DATA, "mcp", "mcp_n30_p0.1_000.mcp"), "-w", "100"])
    assert rc == 0 and kv["Aborted"] == "false" and kv["Gap"] == "0.000"
    assert kv["Solution"].startswith("[Decision { variable: Variable(") and kv["Solution"].count("Decision") == 30


@pytest.mark.gpu
def test_tsptw_report():
    """examples/tsptw/main.rs:102-146: six lines, bounds as tour lengths with two decimals, the tour as a permutation"""
    rc, kv, lines = run(["tsptw", os.path.join(DATA, "tsptw", "Langevin", "N40ft201.dat"), "-t", "16"])
    assert rc == 0
    kv = {k.strip(): v for k, v in kv.items()}
    assert [ln.split(":")[0].strip() for ln in lines] == ["instance", "status", "lower bnd", "upper bnd", "duration", "solution"]
    assert kv["instance"] == "Langevin/N40ft201.dat" and kv["status"] == "Proved"
    assert kv["lower bnd"] == "1109.30" and kv["upper bnd"] == "1109.30"
    tour = [int(x) for x in kv["solution"].split()]
    assert len(tour) == 40 and tour[-1] == 0 and sorted(tour) == list(range(40))
