"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np


class SolveOut(C.Structure):
    _fields_ = [("has_value", C.c_int), ("is_exact", C.c_int), ("best_value", C.c_int64), ("best_lb", C.c_int64),
                ("best_ub", C.c_int64), ("explored", C.c_uint64), ("nodes_expanded", C.c_uint64), ("arcs", C.c_uint64),
                ("layers", C.c_uint64), ("compiles", C.c_uint64), ("wall_s", C.c_double), ("n_solution", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TraceHdr(C.Structure):
    _fields_ = [("comp_type", C.c_int), ("is_exact", C.c_int), ("has_best", C.c_int), ("has_best_exact", C.c_int),
                ("width", C.c_uint64), ("best_lb", C.c_int64), ("value", C.c_int64), ("ub", C.c_int64),
                ("depth", C.c_uint64), ("best_value", C.c_int64), ("best_exact_value", C.c_int64),
                ("nodes_expanded", C.c_uint64), ("arcs", C.c_uint64), ("layers", C.c_uint64), ("n_cutset", C.c_uint64)]


def _canon(hdr, state, cs_states, cs_value, cs_ub, cs_depth, ws):
    """Canonical, order-independent description of one compile()."""
    cut = sorted((tuple(int(x) for x in cs_states[i * ws:(i + 1) * ws]), int(cs_value[i]), int(cs_ub[i]), int(cs_depth[i]))
                 for i in range(len(cs_value)))
    return {
        "comp_type": hdr.comp_type, "width": hdr.width, "best_lb": hdr.best_lb,
        "state": tuple(int(x) for x in state), "value": hdr.value, "ub": hdr.ub, "depth": hdr.depth,
        "is_exact": bool(hdr.is_exact),
        "best_value": hdr.best_value if hdr.has_best else None,
        "best_exact_value": hdr.best_exact_value if hdr.has_best_exact else None,
        "nodes_expanded": hdr.nodes_expanded, "arcs": hdr.arcs, "layers": hdr.layers,
        "cutset": cut,
    }


def read_trace(L, t, ws):
    """Canonical records of a trace handle (freed here)."""
    recs = []
    try:
        n = L.oracle_trace_len(t)
        for i in range(n):
            hdr = TraceHdr()
            st = np.zeros(ws, dtype=np.uint64)
            L.oracle_trace_get(t, i, C.byref(hdr), st.ctypes.data_as(C.c_void_p))
            k = int(hdr.n_cutset)
            cs = np.zeros(max(k, 1) * ws, dtype=np.uint64)
            cv = np.zeros(max(k, 1), dtype=np.int64)
            cu = np.zeros(max(k, 1), dtype=np.int64)
            cd = np.zeros(max(k, 1), dtype=np.uint64)
            if k:
                L.oracle_trace_get_cutset(t, i, cs.ctypes.data_as(C.c_void_p), cv.ctypes.data_as(C.c_void_p),
                                          cu.ctypes.data_as(C.c_void_p), cd.ctypes.data_as(C.c_void_p))
            recs.append(_canon(hdr, st, cs, cv[:k], cu[:k], cd[:k], ws))
    finally:
        L.oracle_trace_free(t)
    return recs


class MispInstance:
    def __init__(self, oracle, path):
        self.o = oracle
        self.L = oracle.L
        self.h = self.L.oracle_misp_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle could not load {path}")
        self.n = self.L.oracle_misp_nb_vars(self.h)
        self.ws = self.L.oracle_misp_state_words(self.h)
        self.rows = np.zeros(self.n * self.ws, dtype=np.uint64)
        self.weights = np.zeros(self.n, dtype=np.int64)
        self.L.oracle_misp_export(self.h, self.rows.ctypes.data_as(C.c_void_p), self.weights.ctypes.data_as(C.c_void_p))

    def __del__(self):
        try:
            self.L.oracle_misp_free(self.h)
        except Exception:
            pass

    def root_state(self):
        s = np.zeros(self.ws, dtype=np.uint64)
        for i in range(self.n):
            s[i // 64] |= np.uint64(1) << np.uint64(i % 64)
        return s

    def solve(self, width=0, nthreads=0, timeout=0.0, pooled=False):
        """pooled=True: the same search over Pooled DDs (mdd/pooled.rs; solver/mod.rs:34-47)"""
        out = SolveOut()
        sol = np.zeros(2 * self.n + 2, dtype=np.int64)
        fn = self.L.oracle_misp_solve_pooled if pooled else self.L.oracle_misp_solve
        fn(self.h, width, nthreads, timeout, C.byref(out), sol.ctypes.data_as(C.c_void_p))
        d = out.asdict()
        d["solution"] = [(int(sol[2 * i]), int(sol[2 * i + 1])) for i in range(out.n_solution)]
        return d

    def trace_solve(self, width=0, max_compiles=0, pooled=False):
        """Sequential B&B recording every compile(); returns (summary, [canonical records]).  pooled=True: SeqNoCachingSolverPooled."""
        out = SolveOut()
        fn = self.L.oracle_misp_trace_solve_pooled if pooled else self.L.oracle_misp_trace_solve
        t = fn(self.h, width, max_compiles, C.byref(out))
        return out.asdict(), read_trace(self.L, t, self.ws)

    def compile(self, comp_type, width, best_lb, state, value, depth, pooled=False):
        """One compile() of an arbitrary residual sub-problem -> canonical record (+ best path).  pooled=True: as a Pooled DD
        (mdd/pooled.rs; frontier cut-set, which a width does not bound)."""
        hdr = TraceHdr()
        cap = int(width) + 8 if width < (1 << 40) else 1 << 16
        if pooled:
            cap = 1 << 18
        st = np.ascontiguousarray(state, dtype=np.uint64)
        cs = np.zeros(cap * self.ws, dtype=np.uint64)
        cv = np.zeros(cap, dtype=np.int64)
        cu = np.zeros(cap, dtype=np.int64)
        cd = np.zeros(cap, dtype=np.uint64)
        bp = np.zeros(2 * self.n + 2, dtype=np.int64)
        nbp = C.c_int64(0)
        fn = self.L.oracle_misp_compile_pooled if pooled else self.L.oracle_misp_compile
        k = fn(self.h, comp_type, width, best_lb, st.ctypes.data_as(C.c_void_p), value, depth,
                                       C.byref(hdr), cap, cs.ctypes.data_as(C.c_void_p), cv.ctypes.data_as(C.c_void_p),
                                       cu.ctypes.data_as(C.c_void_p), cd.ctypes.data_as(C.c_void_p),
                                       bp.ctypes.data_as(C.c_void_p), C.byref(nbp))
        if k < 0:
            raise RuntimeError(f"oracle_misp_compile failed: {k}")
        rec = _canon(hdr, st, cs, cv[:k], cu[:k], cd[:k], self.ws)
        rec["best_path"] = [(int(bp[2 * i]), int(bp[2 * i + 1])) for i in range(nbp.value)]
        return rec


class Oracle:
    def __init__(self, libpath):
        L = C.CDLL(libpath)
        L.oracle_misp_load.restype = C.c_void_p
        L.oracle_misp_load.argtypes = [C.c_char_p]
        L.oracle_misp_free.argtypes = [C.c_void_p]
        L.oracle_misp_nb_vars.argtypes = [C.c_void_p]
        L.oracle_misp_state_words.argtypes = [C.c_void_p]
        L.oracle_misp_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_misp_solve.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.POINTER(SolveOut), C.c_void_p]
        L.oracle_misp_solve_pooled.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.POINTER(SolveOut), C.c_void_p]
        for fn in (L.oracle_misp_trace_solve, L.oracle_misp_trace_solve_pooled):
            fn.restype = C.c_void_p
            fn.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(SolveOut)]
        L.oracle_trace_free.argtypes = [C.c_void_p]
        L.oracle_trace_len.restype = C.c_uint64
        L.oracle_trace_len.argtypes = [C.c_void_p]
        L.oracle_trace_get.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(TraceHdr), C.c_void_p]
        L.oracle_trace_get_cutset.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for fn in (L.oracle_misp_compile, L.oracle_misp_compile_pooled):
            fn.restype = C.c_int64
            fn.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_void_p, C.c_int64, C.c_uint64,
                           C.POINTER(TraceHdr), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.oracle_knapsack_solve.restype = C.c_int64
        L.oracle_knapsack_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                            C.c_void_p, C.POINTER(SolveOut)]
        L.oracle_knapsack_solve_file.restype = C.c_int64
        L.oracle_knapsack_solve_file.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(SolveOut)]
        L.oracle_max2sat_solve_file.restype = C.c_int64
        L.oracle_max2sat_solve_file.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_double, C.c_void_p, C.POINTER(SolveOut)]
        L.oracle_max2sat_instance_info.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.oracle_max2sat_evaluate.restype = C.c_int64
        L.oracle_max2sat_evaluate.argtypes = [C.c_char_p, C.c_void_p]
        L.oracle_mcp_solve_file.restype = C.c_int64
        L.oracle_mcp_solve_file.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(SolveOut)]
        L.oracle_mcp_cut_weight.restype = C.c_int64
        L.oracle_mcp_cut_weight.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.oracle_tsptw_solve_file.restype = C.c_int64
        L.oracle_tsptw_solve_file.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(SolveOut)]
        L.oracle_tsptw_tour_length.restype = C.c_int64
        L.oracle_tsptw_tour_length.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_uint64)]
        for fn in ("oracle_max2sat_trace_solve", "oracle_mcp_trace_solve"):
            getattr(L, fn).restype = C.c_void_p
            getattr(L, fn).argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(SolveOut)]
        L.oracle_trace_solve_ex.restype = C.c_void_p
        L.oracle_trace_solve_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(SolveOut)]
        L.oracle_trace_state_words.restype = C.c_uint64
        L.oracle_trace_state_words.argtypes = [C.c_void_p]
        self.L = L

    def trace_ex(self, kind, path, width=0, max_compiles=0, frontier=False, cache=False):
        """Traced SEQUENTIAL solve of an instance file ("misp" | "knapsack" | "max2sat" | "mcp") with a last-exact-layer or
        frontier cut-set and the Empty / Simple cache: (summary, canonical records in compile order)."""
        out = SolveOut()
        t = self.L.oracle_trace_solve_ex(kind.encode(), path.encode(), width, max_compiles, 1 if frontier else 0, 1 if cache else 0, C.byref(out))
        if not t:
            raise RuntimeError(f"oracle_trace_solve_ex failed on {kind} {path}")
        ws = int(self.L.oracle_trace_state_words(t))
        return out.asdict(), read_trace(self.L, t, ws)

    def vector_trace(self, kind, path, width=0, max_compiles=0):
        """Traced sequential solve of a signed-vector model ("max2sat" | "mcp"): (summary, canonical records); states
        are packed as on the device wire (two i32 benefits per word + a depth word)."""
        out = SolveOut()
        t = getattr(self.L, f"oracle_{kind}_trace_solve")(path.encode(), width, max_compiles, C.byref(out))
        if not t:
            raise RuntimeError(f"oracle_{kind}_trace_solve failed on {path}")
        ws = int(self.L.oracle_trace_state_words(t))
        return out.asdict(), read_trace(self.L, t, ws)

    def tsptw_file(self, path, width_factor=1, nthreads=1):
        nn = C.c_uint64(0)
        self.L.oracle_tsptw_tour_length(path.encode(), None, C.byref(nn))
        tour = np.full(max(1, nn.value), -1, dtype=np.int64)
        out = SolveOut()
        v = self.L.oracle_tsptw_solve_file(path.encode(), width_factor, nthreads, tour.ctypes.data_as(C.c_void_p), C.byref(out))
        d = out.asdict()
        d["tour"] = [int(x) for x in tour[:nn.value]]
        d["nb_nodes"] = int(nn.value)
        if d["n_solution"]:
            d["tour_length"] = int(self.L.oracle_tsptw_tour_length(path.encode(), tour.ctypes.data_as(C.c_void_p), None))
        return int(v), d

    def mcp_file(self, path, width=0, nthreads=0):
        nv = C.c_uint64(0)
        self.L.oracle_mcp_cut_weight(path.encode(), None, C.byref(nv))
        sol = np.zeros(max(1, nv.value), dtype=np.int64)
        out = SolveOut()
        v = self.L.oracle_mcp_solve_file(path.encode(), width, nthreads, sol.ctypes.data_as(C.c_void_p), C.byref(out))
        d = out.asdict()
        d["solution"] = [int(x) for x in sol[:nv.value]]
        d["nb_vertices"] = int(nv.value)
        if d["n_solution"]:
            d["cut_weight"] = int(self.L.oracle_mcp_cut_weight(path.encode(), sol.ctypes.data_as(C.c_void_p), None))
        return int(v), d

    def max2sat_file(self, path, width=0, nthreads=0, time_budget_s=0.0):
        nv, nc = C.c_uint64(0), C.c_uint64(0)
        assert self.L.oracle_max2sat_instance_info(path.encode(), C.byref(nv), C.byref(nc)) == 0
        sol = np.zeros(max(1, nv.value), dtype=np.int64)
        out = SolveOut()
        v = self.L.oracle_max2sat_solve_file(path.encode(), width, nthreads, time_budget_s, sol.ctypes.data_as(C.c_void_p),
                                             C.byref(out))
        d = out.asdict()
        d["solution"] = [int(x) for x in sol[:nv.value]]
        d["nb_vars"], d["nb_clauses"] = int(nv.value), int(nc.value)
        if d["n_solution"]:
            d["solution_weight"] = int(self.L.oracle_max2sat_evaluate(path.encode(), sol.ctypes.data_as(C.c_void_p)))
        return int(v), d

    def misp(self, path):
        return MispInstance(self, path)

    def knapsack(self, profit, weight, capacity, width=0, nthreads=0):
        p = np.ascontiguousarray(profit, dtype=np.int64)
        w = np.ascontiguousarray(weight, dtype=np.uint64)
        sol = np.zeros(len(p), dtype=np.int64)
        out = SolveOut()
        v = self.L.oracle_knapsack_solve(len(p), p.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), capacity,
                                         width, nthreads, sol.ctypes.data_as(C.c_void_p), C.byref(out))
        d = out.asdict()
        d["solution"] = [int(x) for x in sol]
        return int(v), d

    def knapsack_file(self, path, width=0, nthreads=0):
        out = SolveOut()
        v = self.L.oracle_knapsack_solve_file(path.encode(), width, nthreads, C.byref(out))
        return int(v), out.asdict()
