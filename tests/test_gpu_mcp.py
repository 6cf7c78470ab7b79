"""Maximum cut on the device (`-m gpu`), through the C ABI: the reference's ten known optima (examples/mcp/tests.rs:64-103),
every reported solution re-evaluated as a cut, against the CPU oracle's optimum and bounds.  McpRanking is not a total
order, so explored counts are not comparable: optimum, proof and solution are."""
import numpy as np
import pytest

import ddo_amd
from ddo_amd import FixedWidth, NbUnassignedWidth, ParallelSolver
from tests.conftest import data_path

pytestmark = pytest.mark.gpu

MCP_KAT = [("000", 13), ("001", 18), ("002", 15), ("003", 19), ("004", 16), ("005", 19), ("006", 12), ("007", 18),
           ("008", 20), ("009", 22)]


@pytest.fixture(scope="module")
def have_gpu():
    if ddo_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on an MI355X box")
    return True


def read_graph(path):
    n, adj = 0, None
    for line in open(path):
        t = line.split()
        if not t or line.startswith("c "):
            continue
        if len(t) == 2:
            n = int(t[0])
            adj = np.zeros((n, n), dtype=np.int64)
        elif len(t) == 3:
            a, b, w = int(t[0]) - 1, int(t[1]) - 1, int(t[2])
            adj[a, b] = adj[b, a] = w
    return n, adj


def cut_weight(adj, sides):
    n = len(sides)
    return int(sum(adj[a, b] for a in range(n) for b in range(a + 1, n) if sides[a] * sides[b] < 0))


def check(s, adj, expected):
    c_sol = s.best_solution()
    sides = [0] * adj.shape[0]
    for d in c_sol:
        assert d.value in (1, -1)
        sides[d.variable] = d.value
    assert all(sides) and sides[0] == 1                     # every vertex placed; the first one is fixed on side S
    assert cut_weight(adj, sides) == expected


@pytest.mark.parametrize("idx,expected", MCP_KAT)
@pytest.mark.parametrize("width,threads", [(0, 1), (0, 64), (5, 16), (100, 8)])
def test_mcp_known_optimum(have_gpu, oracle, idx, expected, width, threads):
    path = data_path("mcp", f"mcp_n30_p0.1_{idx}.mcp")
    n, adj = read_graph(path)
    model = ddo_amd.Mcp.read_instance(path)
    assert model.n == n == 30 and model.ws == 16
    assert model.initial_value() == int(adj[adj < 0].sum() // 2) and not model.initial_state().any()
    s = ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(n), nb_threads=threads, fringe="nodup")
    c = s.maximize()
    assert c.is_exact and c.best_value == expected
    assert s.best_lower_bound() == expected and s.best_upper_bound() == expected
    check(s, adj, expected)
    v, _ = oracle.mcp_file(path, width, 0)
    assert v == expected


def test_mcp_negative_weights_from_matrix(have_gpu):
    """a small signed graph: optimum by brute force over the 2^(n-1) cuts"""
    rng = np.random.RandomState(5)
    n = 12
    adj = np.zeros((n, n), dtype=np.int64)
    for a in range(n):
        for b in range(a + 1, n):
            if rng.rand() < 0.5:
                adj[a, b] = adj[b, a] = rng.randint(-9, 10)
    best = max(cut_weight(adj, [1] + [1 if (m >> i) & 1 else -1 for i in range(n - 1)]) for m in range(1 << (n - 1)))
    model = ddo_amd.Mcp.from_matrix(adj)
    for width, threads in ((3, 4), (0, 1), (50, 32)):
        s = ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(n), nb_threads=threads, fringe="nodup")
        c = s.maximize()
        assert c.is_exact and c.best_value == best
        check(s, adj, best)


def test_mcp_wider_than_16_words(have_gpu, oracle, tmp_path):
    """n = 36 needs 19 state words (the 32-word template): same optimum as the CPU oracle, verified cut"""
    rng = np.random.RandomState(11)
    n = 36
    adj = np.zeros((n, n), dtype=np.int64)
    p = tmp_path / "g36.mcp"
    edges = []
    for a in range(n):
        for b in range(a + 1, n):
            if rng.rand() < 0.12:
                w = int(rng.randint(-3, 8)) or 1
                adj[a, b] = adj[b, a] = w
                edges.append((a + 1, b + 1, w))
    p.write_text("c random signed graph\n%d %d\n" % (n, len(edges)) + "".join("%d %d %d\n" % e for e in edges))
    v, info = oracle.mcp_file(str(p), 0, 8)
    assert info["is_exact"] and info["cut_weight"] == v
    model = ddo_amd.Mcp.read_instance(str(p))
    assert model.n == n and model.ws == 19
    s = ParallelSolver(model, NbUnassignedWidth(n), nb_threads=128, fringe="nodup")
    c = s.maximize()
    assert c.is_exact and c.best_value == v
    check(s, adj, v)
