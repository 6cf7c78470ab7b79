"""CPU suite, part 2: the C-ABI library loads, exports every symbol include/ddo_hip.h declares, refuses to
run without a GPU (no CPU fallback), and its host-side logic (parser, ranking) agrees with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

import ddo_amd
from tests.conftest import ROOT, data_path


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ddo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ddo_[a-z_0-9]+)\s*\(", text)) - {"ddo_cutset_cb"})


def test_library_exports_every_declared_symbol():
    L = ddo_amd.lib()
    declared = _declared_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), f"libddo_hip.so does not export {name}"
    assert sorted(ddo_amd.ABI_SYMBOLS) == declared


def test_product_library_does_not_link_the_oracle():
    out = os.popen(f"nm -D --defined-only {ddo_amd.library_path()} | grep -i oracle").read()
    assert out.strip() == ""
    for f in os.listdir(os.path.join(ROOT, "ddo_amd", "csrc")):
        src = open(os.path.join(ROOT, "ddo_amd", "csrc", f), errors="replace").read()
        assert "oracle/" not in src.replace("parity oracle", "") or f.endswith(".md"), f


@pytest.mark.skipif(ddo_amd.device_count() > 0, reason="checks the no-GPU behaviour")
def test_fails_loudly_without_a_gpu():
    model = ddo_amd.Misp.read_instance(data_path("misp", "johnson8-2-4.clq"))
    with pytest.raises(ddo_amd.DdoError, match="no HIP device"):
        ddo_amd.Mdd(model, 10)
    with pytest.raises(ddo_amd.DdoError, match="no HIP device"):
        ddo_amd.ParallelSolver(model, ddo_amd.FixedWidth(10))


@pytest.mark.parametrize("name", ["johnson8-2-4", "MANN_a9", "brock200_2", "p_hat300-1", "brock400_1", "c-fat500-1"])
def test_clq_reader_matches_the_oracle_reader(oracle, name):
    path = data_path("misp", name + ".clq")
    inst = oracle.misp(path)
    model = ddo_amd.Misp.read_instance(path)
    rows, w = model.export()
    assert (model.n, model.ws) == (inst.n, inst.ws)
    assert np.array_equal(rows, inst.rows) and np.array_equal(w, inst.weights)
    assert np.array_equal(model.initial_state(), inst.root_state()) and model.initial_value() == 0
    # rows are complement adjacency rows and keep their diagonal bit (examples/misp/main.rs:280-310)
    for i in (0, model.n // 2, model.n - 1):
        assert (int(rows[i * model.ws + i // 64]) >> (i % 64)) & 1


def test_clq_reader_grammar(tmp_path):
    def load(text):
        p = tmp_path / "x.clq"
        p.write_text(text)
        return ddo_amd.Misp.read_instance(str(p))

    m = load("c a comment\n\n  p edge 3 1  \nn 2 -5\ne 1 3 trailing ok\n")
    rows, w = m.export()
    assert list(w) == [1, -5, 1] and [int(r) for r in rows] == [0b011, 0b111, 0b110]
    for bad in ["p edge 3 1\nx 1 2\n", "c\np edge 3 1\n", "p edge 3 1 extra\n", "e 1 2\n", "p edge 2 1\ne 1 5\n"]:
        with pytest.raises(ddo_amd.DdoError):
            load(bad)


def test_state_ranking_is_popcount_then_member_order():
    """MispRanking (examples/misp/main.rs:205-208) == (len, BitSet::cmp); checked against a naive restatement."""
    m = ddo_amd.Misp.read_instance(data_path("misp", "p_hat300-1.clq"))
    rng = np.random.RandomState(3)

    def members(s):
        return [i for i in range(m.n) if (int(s[i // 64]) >> (i % 64)) & 1]

    for _ in range(300):
        a = rng.randint(0, 1 << 62, size=m.ws).astype(np.uint64) & rng.randint(0, 1 << 62, size=m.ws).astype(np.uint64)
        b = a.copy() if rng.rand() < 0.2 else rng.randint(0, 1 << 62, size=m.ws).astype(np.uint64) & a
        if rng.rand() < 0.5:
            b[rng.randint(m.ws)] ^= np.uint64(1) << np.uint64(rng.randint(60))
        a[-1] &= np.uint64((1 << (m.n % 64)) - 1)
        b[-1] &= np.uint64((1 << (m.n % 64)) - 1)
        ma, mb = members(a), members(b)
        expect = (len(ma) > len(mb)) - (len(ma) < len(mb)) or (ma > mb) - (ma < mb)
        got = m.compare(a, b)
        assert (got > 0) - (got < 0) == expect


def test_wire_structs_match_the_device_header():
    """tests/dd_wire.py mirrors ddo_amd/csrc/dd_types.h; the emulation library reports the C sizes."""
    from tests.dd_wire import DDInput, DDResult
    from tests.emul_binding import build_emul

    L = ctypes.CDLL(build_emul())
    L.emul_sizeof_input.restype = ctypes.c_uint64
    L.emul_sizeof_result.restype = ctypes.c_uint64
    assert L.emul_sizeof_input() == ctypes.sizeof(DDInput) and L.emul_sizeof_result() == ctypes.sizeof(DDResult)


# ---- knapsack model descriptor (examples/knapsack/main.rs:53-194), host side only ------------------------------
def test_knapsack_model_host_side(tmp_path):
    m = ddo_amd.Knapsack.from_items(50, [60, 100, 120], [10, 20, 30])
    assert m.n == 3 and m.ws == 2
    assert [int(x) for x in m.initial_state()] == [50, 0] and m.initial_value() == 0
    # KPRanking (main.rs:187-194): the remaining capacity alone, whatever the depth word says
    a, b = np.array([7, 3], dtype=np.uint64), np.array([9, 1], dtype=np.uint64)
    assert m.compare(a, b) < 0 and m.compare(b, a) > 0 and m.compare(a, np.array([7, 5], dtype=np.uint64)) == 0
    # reader (main.rs:267-303): comment lines, "n capacity", n x "profit weight", anything after the n items ignored
    p = tmp_path / "kp.txt"
    p.write_text("c a comment\n3 50\n60 10\nc another\n100 20\n120 30\n999 1\n")
    r = ddo_amd.Knapsack.read_instance(p)
    assert r.n == 3 and int(r.initial_state()[0]) == 50
    for bad in ("3 50\n60 10\n", "", "c only comments\n"):
        q = tmp_path / "bad.txt"
        q.write_text(bad)
        with pytest.raises(ddo_amd.DdoError):
            ddo_amd.Knapsack.read_instance(q)
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Knapsack.from_items(10, [1, 2], [0, 3])       # weights must be >= 1
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Knapsack.from_items(-1, [1], [1])


@pytest.mark.parametrize("name", ["f1_l-d_kp_10_269", "f8_l-d_kp_23_10000", "knapPI_1_100_1000_1"])
def test_knapsack_reader_agrees_with_the_oracle(oracle, name):
    """same optimum from the oracle whether it parses the file itself or gets the product reader's arrays"""
    path = data_path("knapsack", name)
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("c")]
    n, cap = int(rows[0][0]), int(rows[0][1])
    m = ddo_amd.Knapsack.read_instance(path)
    assert m.n == n and int(m.initial_state()[0]) == cap
    if n <= 23:
        v_file, _ = oracle.knapsack_file(path, 10, 0)
        v_arr, _ = oracle.knapsack([int(r[0]) for r in rows[1:1 + n]], [int(r[1]) for r in rows[1:1 + n]], cap, 10, 0)
        assert v_file == v_arr


# ---- MAX2SAT / MCP model descriptors, host side only -----------------------------------------------------------------
def test_max2sat_model_host_side(tmp_path):
    """data.rs:67-126: debug2.wcnf has 3 variables; unit clause written `w x x 0`; a repeated clause keeps its last weight"""
    m = ddo_amd.Max2Sat.read_instance(data_path("max2sat", "debug2.wcnf"))
    assert m.n == 3 and m.ws == 3 and not m.initial_state().any() and m.initial_value() == 0
    p = tmp_path / "t.wcnf"
    p.write_text("c comment\np wcnf 2 3\n5 1 -1 0\n3 1 2 0\n7 1 2 0\n4 -2 0\n")   # tautology 5, (1 v 2) re-weighted to 7, unit -2
    t = ddo_amd.Max2Sat.read_instance(p)
    assert t.n == 2 and t.initial_value() == 5
    same = ddo_amd.Max2Sat.from_clauses(2, [(1, -1, 5), (2, 1, 7), (-2, -2, 4)])
    assert same.initial_value() == 5
    # Max2SatRanking (heuristics.rs:30-37): sum of |benefit|; ties fall to the packed words (the deterministic tie-break
    # shared by oracle, host and device: SURVEY.md section 7), word 0 first, the depth word last
    def pack(vals, depth):
        w = np.zeros(m.ws, dtype=np.uint64)
        for i, v in enumerate(vals):
            w[i // 2] |= np.uint64(v & 0xFFFFFFFF) << np.uint64(32 * (i % 2))
        w[(len(vals) + 1) // 2] = np.uint64(depth)
        return w
    assert m.compare(pack([1, -2, 0], 1), pack([0, 0, 4], 2)) < 0 and m.compare(pack([3, -3, 0], 0), pack([-6, 0, 0], 5)) > 0
    assert m.compare(pack([3, -3, 0], 0), pack([3, -3, 0], 5)) < 0 and m.compare(pack([3, -3, 0], 5), pack([3, -3, 0], 5)) == 0
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Max2Sat.from_clauses(2, [(3, 1, 1)])            # literal outside [-n, n]
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Max2Sat.from_clauses(143, [])                     # more than 142 variables


def test_mcp_model_host_side(tmp_path):
    m = ddo_amd.Mcp.read_instance(data_path("mcp", "mcp_n30_p0.1_000.mcp"))
    assert m.n == 30 and m.ws == 16 and not m.initial_state().any()
    p = tmp_path / "g.mcp"
    p.write_text("c tiny\n3 2\n1 2 -4\n2 3 5\n")
    g = ddo_amd.Mcp.read_instance(p)
    assert g.n == 3 and g.initial_value() == -4              # graph.rs:37-42: sum of the negative edges
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Mcp.from_matrix(np.array([[0, 1], [2, 0]]))    # not symmetric
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Mcp.from_matrix(np.zeros((143, 143), dtype=np.int64))    # more than 142 vertices


def test_width_heuristics_known_answers():
    """width.rs:937-1075 (test_fixedwidth, test_adapters) and the doc examples width.rs:492-505, 732-745: Times / DivBy decorate
    any inner heuristic; both floor at 1."""
    from ddo_amd import DivBy, FixedWidth, NbUnassignedWidth, Times, TsptwWidth, width_heuristic as w
    for depth in (0, 1, 5):
        assert w(FixedWidth(5), 5, depth) == 5                                   # width.rs:954, 966, 985
    assert [w(Times(k, FixedWidth(5)), 9, 1) for k in (2, 3, 1, 10)] == [10, 15, 5, 50]   # width.rs:1011-1014
    assert w(DivBy(2, FixedWidth(4)), 9, 1) == 2 and w(DivBy(3, FixedWidth(9)), 9, 1) == 3 and w(DivBy(1, FixedWidth(10)), 9, 1) == 10   # :1032-1034
    assert w(Times(0, FixedWidth(10)), 9, 1) == 1                                 # width.rs:1053: never below 1
    assert w(DivBy(9, FixedWidth(4)), 9, 1) == 1
    assert w(NbUnassignedWidth(5), 5, 1) == 4 and w(NbUnassignedWidth(5), 5, 0) == 5   # width.rs:901, 913
    assert w(Times(5, NbUnassignedWidth(5)), 5, 3) == 10                         # width.rs:492-505 (three of five variables decided)
    assert w(DivBy(2, NbUnassignedWidth(5)), 5, 3) == 1                          # width.rs:732-745
    assert w(DivBy(4, Times(3, NbUnassignedWidth(10))), 10, 2) == 6
    assert w(TsptwWidth(2), 7, 3) == 7 * 4 * 2                                   # tsptw/heuristics.rs:48-52
    # the decorators see the inner policy's RAW value: at depth == nb_vars NbUnassignedWidth is 0 and Times(k, .) is max(1, k * 0) = 1
    # (the inner value used to be clamped to 1 first, which made this k); the engine itself never gets less than 1
    assert w(Times(7, NbUnassignedWidth(5)), 5, 5) == 1 and w(NbUnassignedWidth(5), 5, 5) == 1
    assert w(DivBy(2, Times(4, NbUnassignedWidth(6))), 6, 6) == 1
    import pytest
    with pytest.raises(ZeroDivisionError):                                       # width.rs:1073: DivBy(0, ..) panics
        DivBy(0, FixedWidth(3))


def test_solver_aliases_of_the_reference():
    """solver/mod.rs:29-47: fourteen aliases, (parallel | sequential) x (LEL | FC | Pooled) x (EmptyCache | SimpleCache), plus the two
    defaults.  All of them exist on the device under the reference's names (round 5: `Pooled` and the two NoCaching Pooled solvers,
    round 6: Pooled behind a SimpleCache -- their GPU tests are tests/test_gpu_pooled.py)."""
    built = ["DefaultSolver", "DefaultCachingSolver", "ParNoCachingSolverLel", "ParNoCachingSolverFc", "ParCachingSolverLel", "ParCachingSolverFc",
             "SeqNoCachingSolverLel", "SeqNoCachingSolverFc", "SeqCachingSolverLel", "SeqCachingSolverFc", "Pooled", "ParNoCachingSolverPooled",
             "SeqNoCachingSolverPooled", "ParCachingSolverPooled", "SeqCachingSolverPooled"]
    for name in built:
        assert callable(getattr(ddo_amd, name)), name
