"""Parity tests proper (`-m gpu`): the HIP engine, called through the C ABI, against the CPU oracle
on the same seeded inputs, against the committed golden fixtures, and through size-independent
properties at BASELINE.json's full sizes.  Bit-exact: all arithmetic is integer."""
import json
import os

import numpy as np
import pytest

import ddo_amd
from ddo_amd import CompilationType, FixedWidth, NbUnassignedWidth, ParallelSolver, SubProblem
from tests.conftest import data_path
from tests.parity_util import ENGINES, KEYS, canon_from_mdd, check_replay, cutset_digest, diff, engine_width, is_independent_set, may_hand_up, replay_records

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "misp_compile_golden.json")


@pytest.fixture(scope="module")
def have_gpu():
    if ddo_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on an MI355X box")
    return True


# ---- (1) every compile() of a sequential oracle B&B replayed on the GPU ---------------------------
@pytest.mark.parametrize("name,width,max_compiles", [
    ("johnson8-4-4", 0, 0),        # NbUnassignedWidth (examples/misp/tests.rs), whole search
    ("MANN_a9", 0, 0),
    ("hamming6-4", 0, 0),
    ("johnson8-2-4", 0, 0),
    ("brock200_2", 0, 600),
    ("brock200_2", 1000, 0),       # BASELINE config C2: width 1000, whole search
    ("brock200_2", 1, 200),        # width 1: everything merges into one node per layer
    ("brock200_2", 2, 200),
    ("brock200_2", 37, 300),
    ("keller4", 0, 400),
    ("hamming8-4", 0, 200),        # n = 256: exactly 4 words
    ("p_hat300-1", 0, 300),        # n = 300: 5 words padded to 7
    ("c-fat500-1", 0, 0),          # n = 500: 8 words
    ("brock400_1", 500, 120),      # n = 400: 7 words
    ("keller4", 3000, 500),        # widths far above the capacity tiers' layer capacity: the tiers complete most compiles
    ("p_hat300-1", 5000, 400),
])
@pytest.mark.parametrize("engine", ENGINES)
def test_replay_of_oracle_trace(have_gpu, oracle, name, width, max_compiles, engine):
    """`engine` binds the mdds to one kernel of the in-place engine (DDO_MDD_ENGINE_*): the full-width kernel, the dense
    kernel bench.py times (512 threads, 8-bit select digits, 128 tie-break keys in LDS, half-size table) and the two capacity
    tiers -- every one of them is compared with the oracle compile by compile, on the GPU."""
    path = data_path("misp", name + ".clq")
    inst = oracle.misp(path)
    _, recs = inst.trace_solve(width, max_compiles)
    assert recs
    model = ddo_amd.Misp.read_instance(path)
    tier = engine in ("tier0", "tier1")
    completed, handed = check_replay(model, recs, engine, f"{name} W={width}", min_completed=0 if tier else 1)
    if not tier:
        assert handed == 0
    elif width >= 1000 and max_compiles == 0:   # a whole wide search is made of narrow DDs deep down: those a capacity tier completes
        assert completed >= 20, (completed, handed)
    print(f"[{engine}] {name} W={width}: {completed} compiles equal the oracle's, {handed} handed up")


@pytest.mark.parametrize("name,width,table,max_compiles", [
    ("brock200_2", 100, 128, 400), ("keller4", 100, 128, 300), ("brock200_2", 1000, 2048, 60), ("p_hat300-1", 128, 256, 300),
])
def test_replay_through_a_shrunk_dense_table(have_gpu, oracle, monkeypatch, name, width, table, max_compiles):
    """The dense kernel with a dedup table far smaller than 3 x the layer capacity (DDO_HIP_DENSE_TABLE): layers whose nodes
    plus YES-children would fill more than 7/8 of it hand the DD up (DDO_HANDED_UP), everything it completes is compared
    with the oracle compile by compile -- long probe chains, a nearly full table."""
    monkeypatch.setenv("DDO_HIP_DENSE_TABLE", str(table))
    path = data_path("misp", name + ".clq")
    _, recs = oracle.misp(path).trace_solve(width, max_compiles)
    model = ddo_amd.Misp.read_instance(path)
    completed = handed = 0
    for i, r, got in replay_records(model, recs, engine="dense"):
        if got is ddo_amd.HANDED_UP:
            assert r["nodes_expanded"] > table * 7 // 16, f"{name} compile #{i}: handed up with {r['nodes_expanded']} nodes in all"
            handed += 1
            continue
        d = diff(r, got)
        assert d is None, f"{name} W={width} table={table} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        completed += 1
    assert completed > 0 and (handed > 0 or table > 128), (completed, handed)
    print(f"[dense, table {table}] {name} W={width}: {completed} compiles equal the oracle's, {handed} handed up")


# ---- (2) whole searches: same optimum / proof / explored count as the oracle -----------------------
@pytest.mark.parametrize("name,expected", [
    ("brock200_2", 12), ("johnson8-4-4", 14), ("MANN_a9", 16), ("hamming6-2", 32), ("hamming6-4", 4),
    ("johnson8-2-4", 4), ("c-fat200-1", 12), ("c-fat200-2", 24), ("c-fat200-5", 58), ("p_hat300-1", 8),
    ("keller4", 11), ("c-fat500-1", 14), ("c-fat500-2", 26), ("hamming8-2", 128),
])
def test_solver_known_optimum_sequential_parity(have_gpu, oracle, name, expected):
    """examples/misp/tests.rs:71-161 known answers; nb_threads = 1 must also reproduce the oracle's
    ParallelSolver(1 thread) search exactly (MISP's ranking is a total order)."""
    path = data_path("misp", name + ".clq")
    model = ddo_amd.Misp.read_instance(path)
    s = ParallelSolver(model, NbUnassignedWidth(model.n), nb_threads=1)
    c = s.maximize()
    assert c.is_exact and c.best_value == expected
    assert s.best_lower_bound() == expected and s.best_upper_bound() == expected and s.gap() == 0.0
    sol = s.best_solution()
    chosen = [d.variable for d in sol if d.value == 1]
    rows, w = model.export()
    assert is_independent_set(rows, model.ws, chosen)
    assert sum(int(w[v]) for v in chosen) == expected
    ref = oracle.misp(path).solve(0, 1)
    assert ref["best_value"] == expected
    assert s.explored() == ref["explored"]
    cnt = s.counters()
    assert (cnt["nodes_expanded"], cnt["arcs"], cnt["layers"], cnt["compiles"]) == \
           (ref["nodes_expanded"], ref["arcs"], ref["layers"], ref["compiles"])


@pytest.mark.parametrize("name,expected,width,threads", [
    ("brock200_2", 12, 1000, 64), ("brock200_3", 15, 0, 256), ("brock200_4", 17, 200, 128), ("keller4", 11, 50, 512),
    ("hamming8-4", 16, 0, 256),
])
def test_solver_known_optimum_concurrent(have_gpu, name, expected, width, threads):
    """Many sub-problems in flight (== the reference with that many threads): same proved optimum."""
    model = ddo_amd.Misp.read_instance(data_path("misp", name + ".clq"))
    s = ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), nb_threads=threads)
    c = s.maximize()
    assert c.is_exact and c.best_value == expected
    assert s.best_upper_bound() == expected
    chosen = [d.variable for d in s.best_solution() if d.value == 1]
    rows, _ = model.export()
    assert len(chosen) == expected and is_independent_set(rows, model.ws, chosen)


@pytest.mark.parametrize("name,expected,width,threads", [
    ("brock200_2", 12, 1000, 64), ("brock200_2", 12, 0, 1), ("brock200_3", 15, 0, 256), ("keller4", 11, 50, 300),
    ("johnson8-4-4", 14, 0, 7), ("MANN_a9", 16, 3, 32), ("p_hat300-1", 8, 0, 128), ("c-fat500-1", 14, 0, 8),
])
def test_solver_lazy_device_fringe(have_gpu, name, expected, width, threads):
    """DDO_FRINGE_LAZY: cut-sets never leave HBM (device node pool), the host orders blocks by (ub, value) -- the
    SimpleFringe/MaxUB configuration of the reference (fringe/simple.rs:35-62).  Same proved optimum, and the
    incumbent's path (rebuilt from the pool's bit strings) is a feasible solution of that value."""
    model = ddo_amd.Misp.read_instance(data_path("misp", name + ".clq"))
    s = ParallelSolver(model, FixedWidth(width) if width else NbUnassignedWidth(model.n), nb_threads=threads, fringe="lazy")
    c = s.maximize()
    assert c.is_exact and c.best_value == expected
    assert s.best_upper_bound() == expected and s.fringe_len() == 0
    sol = s.best_solution()
    assert sorted(d.variable for d in sol) == sorted(set(d.variable for d in sol))   # each variable decided once
    chosen = [d.variable for d in sol if d.value == 1]
    rows, w = model.export()
    assert sum(int(w[v]) for v in chosen) == expected and is_independent_set(rows, model.ws, chosen)


@pytest.mark.parametrize("name,expected,width,world", [
    ("brock200_2", 12, 100, 2), ("brock200_2", 12, 30, 3), ("keller4", 11, 20, 2), ("p_hat300-1", 8, 25, 4),
    ("johnson8-4-4", 14, 8, 3), ("hamming6-4", 4, 6, 2),
])
def test_sharded_lazy_fringe(have_gpu, name, expected, width, world):
    """The root cut-set is dealt by hash(state) % world (row positions are scheduling dependent and must not decide):
    unit-weight instances are full of (ub, value) ties, where a position-based deal lost or duplicated sub-problems.
    Every shard is searched to exhaustion; together they prove the optimum, and with the incumbent exchange switched
    OFF the shards still cover the problem (each open node belongs to exactly one rank)."""
    model = ddo_amd.Misp.read_instance(data_path("misp", name + ".clq"))
    for exchange in (True, False):
        ranks = [ParallelSolver(model, FixedWidth(width), nb_threads=16, rank=r, world_size=world, fringe="lazy") for r in range(world)]
        live = [True] * world
        while any(live):
            for r, s in enumerate(ranks):
                if live[r]:
                    live[r] = s.step() == 1
            if exchange:
                lb = max(s.best_lower_bound() for s in ranks)
                for s in ranks:
                    s.import_lower_bound(lb)
        for s in ranks:
            s.flush()
        assert max(s.best_lower_bound() for s in ranks) == expected, (name, width, world, exchange)
        assert all(s.fringe_len() == 0 for s in ranks)


@pytest.mark.parametrize("name,expected,width,tiers,threads", [
    ("brock200_2", 12, 100, "8:64,32:64", 1), ("brock200_2", 12, 100, "16:64", 48), ("keller4", 11, 64, "8:64,24:128", 1),
    ("p_hat300-1", 8, 128, "32:64,64:256", 1), ("brock200_4", 17, 300, "16:64,100:256", 256), ("MANN_a9", 16, 40, "8:64", 1),
])
def test_capacity_tiers_do_not_change_the_search(have_gpu, monkeypatch, name, expected, width, tiers, threads):
    """Capacity tiers (host_solver.cpp: dispatch): narrow DDs are compiled by engines with few node slots per DD and many
    DDs per CU; a DD that outgrows a tier is compiled again by the next one.  A tier never squashes, so whatever it
    completes is what the full-width engine produces (tests/test_emulation.py replays oracle traces through a tier
    configuration compile by compile).  The lazy fringe orders equal (ub, value) rows of a cut-set block by their row
    index, which device atomics decide, so two runs of the same search may pop ties in a different order: the proved
    optimum and a feasible solution of that value are what must not change; the amount of work stays within a few
    percent."""
    model = ddo_amd.Misp.read_instance(data_path("misp", name + ".clq"))

    def run(spec):
        monkeypatch.setenv("DDO_HIP_TIERS", spec)
        s = ParallelSolver(model, FixedWidth(width), nb_threads=threads, fringe="lazy")
        c = s.maximize()
        assert c.is_exact and c.best_value == expected
        sol = [d.variable for d in s.best_solution() if d.value == 1]
        rows, w = model.export()
        assert len(sol) == expected and is_independent_set(rows, model.ws, sol)
        return s.explored(), s.counters()

    base = run("0")
    tiered = run(tiers)
    assert tiered[1]["compiles"] > 0
    if threads == 1:
        assert abs(tiered[0] - base[0]) <= 0.2 * base[0] + 8


@pytest.mark.parametrize("name,expected,width,table,threads", [
    ("brock200_2", 12, 100, 128, 1), ("brock200_2", 12, 100, 1 << 20, 64), ("keller4", 11, 100, 128, 1),
    ("p_hat300-1", 8, 128, 256, 300), ("brock200_4", 17, 300, 512, 256), ("brock200_2", 12, 3000, 4096, 600),
])
def test_the_dense_tier_does_not_change_the_search(have_gpu, monkeypatch, name, expected, width, table, threads):
    """The dense tier (two full-width decision diagrams per CU, half-size dedup table; host_solver.cpp: ddo_solver_create,
    Engine::create_tier) hands a DD whose table could overflow to the full-width engine (ST_RETRY).  Whatever it completes
    is what the full-width engine produces (tests/test_emulation.py replays oracle traces through the dense configuration
    compile by compile); here: same proved optimum, a feasible solution of that value, about the same amount of work."""
    model = ddo_amd.Misp.read_instance(data_path("misp", name + ".clq"))
    monkeypatch.setenv("DDO_HIP_TIERS", "0")

    def run(dense):
        monkeypatch.setenv("DDO_HIP_DENSE", "1" if dense else "0")
        monkeypatch.setenv("DDO_HIP_DENSE_TABLE", str(table))
        s = ParallelSolver(model, FixedWidth(width), nb_threads=threads, fringe="lazy")
        c = s.maximize()
        assert c.is_exact and c.best_value == expected
        sol = [d.variable for d in s.best_solution() if d.value == 1]
        rows, w = model.export()
        assert len(sol) == expected and is_independent_set(rows, model.ws, sol)
        return s.explored(), s.tier_stats()

    base = run(False)
    dense = run(True)
    assert len(base[1]) == 1 and len(dense[1]) == 2 and dense[1][0]["dense"] == 1 and dense[1][0]["threads"] == 512
    assert dense[1][0]["subproblems"] > 0 and dense[1][0]["nodes_expanded"] > 0
    if table == 128:   # a table this small overflows on most wide layers
        assert dense[1][0]["retried"] > 0 and dense[1][1]["subproblems"] >= dense[1][0]["retried"]   # (depths that always outgrow the table start on the full engine)
    if threads == 1:
        assert abs(dense[0] - base[0]) <= 0.2 * base[0] + 8


# ---- (3) golden fixtures (generated by tests/golden/make_golden.py from the oracle) ----------------
def _golden_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["id"])
def test_golden_compile(have_gpu, case, engine):
    model = ddo_amd.Misp.read_instance(data_path("misp", case["instance"] + ".clq"))
    mdd = ddo_amd.Mdd(model, engine_width(engine, case["width"]), engine=engine)
    state = np.array([int(x) for x in case["state"]], dtype=np.uint64)
    sub = SubProblem(state=state, value=case["value"], path=[], depth=case["depth"])
    comp = mdd.compile(case["comp_type"], case["width"], sub, case["best_lb"])
    if comp is ddo_amd.HANDED_UP:
        assert may_hand_up(engine, case), f"{case['id']} [{engine}]: handed up although it fits the tier"
        return
    got = canon_from_mdd(mdd, comp, model.ws)
    for k in ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers"]:
        assert got[k] == case[k], f"{case['id']} [{engine}]: {k} expected {case[k]} got {got[k]}"
    assert len(got["cutset"]) == case["n_cutset"]
    assert cutset_digest(got["cutset"]) == case["cutset_digest"]
    # paths: every cut-set node's path replays to its state and value (clean.rs:430-441)
    rows, w = model.export()
    for node in got["cutset_nodes"][:64]:
        s = state.copy()
        v = case["value"]
        assert len(node.path) == node.depth - case["depth"]
        for d in reversed(node.path):   # node-first order: replay from the root side
            bit = np.uint64(1) << np.uint64(d.variable % 64)
            has = bool(s[d.variable // 64] & bit)
            assert has or d.value == 0
            s[d.variable // 64] &= ~bit
            if d.value == 1:
                s &= rows[d.variable * model.ws:(d.variable + 1) * model.ws]
                v += int(w[d.variable])
        assert tuple(int(x) for x in s) == tuple(int(x) for x in node.state) and v == node.value


# ---- (4) full-size properties (BASELINE config C4: brock400_1, width 10 000) ------------------------
def test_full_size_properties_brock400_w10000(have_gpu):
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock400_1.clq"))
    rows, w = model.export()
    W = 10000
    mdd = ddo_amd.Mdd(model, W)
    root = model.root()
    lb = -(1 << 40)
    r = mdd.compile(CompilationType.Restricted, W, root, lb)
    assert r is not None and not r.is_exact and r.best_value is not None
    sol = mdd.best_solution()
    chosen = [d.variable for d in sol if d.value == 1]
    assert is_independent_set(rows, model.ws, chosen) and len(chosen) == r.best_value   # feasible lower bound
    assert mdd.best_exact_value() == r.best_value
    restricted_nodes = mdd.counters()["nodes_expanded"]
    x = mdd.compile(CompilationType.Relaxed, W, root, r.best_value)
    assert x is not None and not x.is_exact
    assert x.best_value >= 27 >= r.best_value          # relaxation bounds the (literature) optimum 27 from above
    cut = mdd.drain_cutset()
    assert 0 < len(cut) <= W
    assert mdd.drain_cutset() == []                     # drained at most once (mdd.rs:107-110)
    depth = cut[0].depth
    seen = set()
    for n in cut:
        assert n.depth == depth and len(n.path) == depth
        assert r.best_value < n.ub <= x.best_value      # device-side filter semantics are off here: all marked nodes
        assert n.value <= n.ub
        key = tuple(int(v) for v in n.state)
        assert key not in seen                          # an exact layer holds each state once
        seen.add(key)
        assert n.value == sum(d.value for d in n.path)  # unit weights
    # idempotence: compiling again gives the very same observable result
    c1 = sorted((tuple(int(v) for v in n.state), n.value, n.ub) for n in cut)
    x2 = mdd.compile(CompilationType.Relaxed, W, root, r.best_value)
    c2 = sorted((tuple(int(v) for v in n.state), n.value, n.ub) for n in mdd.drain_cutset())
    assert x2.best_value == x.best_value and c1 == c2
    assert mdd.counters()["nodes_expanded"] >= restricted_nodes // 2


# ---- (5) edge cases ---------------------------------------------------------------------------------
def _clique_free_model(n, edges, weights=None):
    ws = (n + 63) // 64
    rows = np.zeros(n * ws, dtype=np.uint64)
    for i in range(n):
        for j in range(n):
            rows[i * ws + j // 64] |= np.uint64(1) << np.uint64(j % 64)
    for a, b in edges:
        rows[a * ws + b // 64] &= ~(np.uint64(1) << np.uint64(b % 64))
        rows[b * ws + a // 64] &= ~(np.uint64(1) << np.uint64(a % 64))
    return ddo_amd.Misp.from_rows(n, rows, weights if weights is not None else np.ones(n, dtype=np.int64))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 129])
def test_edgeless_and_ragged_sizes(have_gpu, n):
    """No edges: the optimum takes every vertex; exercises ragged last words and 1..3-word states."""
    model = _clique_free_model(n, [])
    s = ParallelSolver(model, FixedWidth(3), nb_threads=4)
    c = s.maximize()
    assert c.is_exact and c.best_value == n
    assert sorted(d.variable for d in s.best_solution() if d.value == 1) == list(range(n))


def test_weighted_with_negative_weights(have_gpu, oracle, tmp_path):
    """`n` lines (main.rs:290-297), including a negative weight: values, RUB and ranking use the weights."""
    rng = np.random.RandomState(7)
    n = 40
    edges = [(a, b) for a in range(n) for b in range(a + 1, n) if rng.rand() < 0.3]
    weights = rng.randint(-3, 20, size=n)
    p = tmp_path / "weighted.clq"
    with open(p, "w") as f:
        f.write("c weighted test instance\n")
        f.write(f"p edge {n} {len(edges)}\n")
        for i, wv in enumerate(weights):
            f.write(f"n {i + 1} {int(wv)}\n")
        for a, b in edges:
            f.write(f"e {a + 1} {b + 1}\n")
    inst = oracle.misp(str(p))
    model = ddo_amd.Misp.read_instance(str(p))
    rows, w = model.export()
    assert np.array_equal(rows, inst.rows) and np.array_equal(w, inst.weights)
    for width in (0, 5):
        _, recs = inst.trace_solve(width, 0)
        for i, r, got in replay_records(model, recs):
            assert diff(r, got) is None, (width, i, diff(r, got))
    s = ParallelSolver(model, FixedWidth(8), nb_threads=16)
    assert s.maximize().best_value == inst.solve(8, 0)["best_value"]


def test_everything_pruned_by_best_lb(have_gpu):
    """clean.rs:1670-1749: with best_lb above every bound nothing is expanded and there is no solution."""
    model = ddo_amd.Misp.read_instance(data_path("misp", "johnson8-2-4.clq"))
    mdd = ddo_amd.Mdd(model, 10)
    for t in (CompilationType.Exact, CompilationType.Restricted, CompilationType.Relaxed):
        c = mdd.compile(t, 10, model.root(), 1000)
        assert c is not None and c.best_value is None
        assert mdd.best_solution() is None and mdd.best_value() is None and mdd.drain_cutset() == []


def test_exact_compile_and_capacity_error(have_gpu, oracle):
    """Exact DDs (mdd.rs:43) unroll completely when they fit the workspace, else fail loudly."""
    path = data_path("misp", "johnson8-2-4.clq")
    model = ddo_amd.Misp.read_instance(path)
    mdd = ddo_amd.Mdd(model, 4096)
    c = mdd.compile(CompilationType.Exact, 4096, model.root(), -(1 << 40))
    ref = oracle.misp(path).compile(0, 1 << 40, -(1 << 40), model.initial_state(), 0, 0)
    assert c.is_exact and c.best_value == ref["best_value"] == 4
    assert mdd.counters()["nodes_expanded"] == ref["nodes_expanded"]
    small = ddo_amd.Mdd(model, 4)
    with pytest.raises(ddo_amd.DdoError):
        small.compile(CompilationType.Exact, 4, model.root(), -(1 << 40))
    with pytest.raises(ddo_amd.DdoError):
        small.compile(CompilationType.Restricted, 5, model.root(), 0)   # wider than the mdd was created for


def test_sharded_search_covers_the_problem(have_gpu):
    """Two ranks, each owning half of the root cut-set and exchanging only the incumbent (what bench.py does
    over RCCL): together they prove the same optimum."""
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock200_2.clq"))
    ranks = [ParallelSolver(model, FixedWidth(100), nb_threads=32, rank=r, world_size=2) for r in range(2)]
    live = [True, True]
    while any(live):
        for r, s in enumerate(ranks):
            if live[r]:
                live[r] = s.step() == 1
        lb = max(s.best_lower_bound() for s in ranks)
        for s in ranks:
            s.import_lower_bound(lb)
    assert max(s.best_lower_bound() for s in ranks) == 12
    assert any(s.best_value() == 12 for s in ranks)


# ---- (6) maximum sizes: the 16-word state template (n = 600 pads 10 words to 16; n = 1024 is the largest model) ------
def _write_random_clq(path, n, p_edge, seed, weights=None):
    rng = np.random.RandomState(seed)
    with open(path, "w") as f:
        edges = []
        for a in range(n):
            nb = np.nonzero(rng.rand(n - a - 1) < p_edge)[0]
            edges.extend((a, a + 1 + int(b)) for b in nb)
        f.write(f"p edge {n} {len(edges)}\n")
        if weights is not None:
            for i, wv in enumerate(weights):
                f.write(f"n {i + 1} {int(wv)}\n")
        for a, b in edges:
            f.write(f"e {a + 1} {b + 1}\n")


@pytest.mark.parametrize("n,p_edge,width,max_compiles,weighted", [
    (600, 0.05, 40, 12, False),      # sparse complement graph: dense conflict structure, deep DDs
    (1024, 0.5, 64, 10, False),      # the largest model the ABI accepts (MAX_WS = 16 words)
    (1024, 0.9, 300, 6, True),       # weighted: values, rough upper bounds and ranking use the weights
])
@pytest.mark.parametrize("engine", ENGINES)
def test_replay_at_maximum_state_sizes(have_gpu, oracle, tmp_path, n, p_edge, width, max_compiles, weighted, engine):
    p = tmp_path / f"rand{n}.clq"
    weights = np.random.RandomState(n).randint(1, 50, size=n) if weighted else None
    _write_random_clq(p, n, p_edge, seed=n + width, weights=weights)
    inst = oracle.misp(str(p))
    model = ddo_amd.Misp.read_instance(str(p))
    assert model.n == n and model.ws == (n + 63) // 64
    rows, w = model.export()
    assert np.array_equal(rows, inst.rows) and np.array_equal(w, inst.weights)
    _, recs = inst.trace_solve(width, max_compiles)
    assert recs
    check_replay(model, recs, engine, f"n={n} W={width}", min_completed=0 if engine in ("tier0", "tier1") else 1)


@pytest.mark.parametrize("fringe", ["nodup", "lazy"])
def test_time_budget_aborts_the_search(fringe):
    """TimeBudget (cutoff.rs:302-323) -> abort_search (parallel.rs:479-489): maximize() returns within about the
    budget, not exact, with the incumbent as the lower bound and the best open bound above it."""
    import time
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock400_1.clq"))
    s = ddo_amd.ParallelSolver(model, ddo_amd.FixedWidth(1000), ddo_amd.TimeBudget(1.0), nb_threads=256, fringe=fringe)
    t0 = time.perf_counter()
    c = s.maximize()
    dt = time.perf_counter() - t0
    assert dt < 20.0
    assert not c.is_exact and c.best_value is not None and c.best_value >= 15
    assert s.best_lower_bound() == c.best_value
    assert s.best_upper_bound() >= s.best_lower_bound()
    assert s.gap() > 0.0
    sol = sorted(d.variable for d in s.best_solution() if d.value == 1)
    assert len(sol) == c.best_value
    assert s.step() == ddo_amd.binding.DDO_CUTOFF   # an aborted solver stays aborted
