"""ctypes binding of include/ddo_hip.h.  Names follow the reference crate (ddo/src/lib.rs:497-503)."""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# every symbol include/ddo_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "ddo_last_error", "ddo_device_count", "ddo_model_create_misp", "ddo_model_read_misp", "ddo_model_create_knapsack",
    "ddo_model_read_knapsack", "ddo_model_create_mcp", "ddo_model_read_mcp", "ddo_model_create_max2sat",
    "ddo_model_read_max2sat", "ddo_model_create_tsptw", "ddo_model_read_tsptw", "ddo_model_destroy",
    "ddo_model_nb_variables", "ddo_model_state_words", "ddo_model_initial_state", "ddo_model_initial_value",
    "ddo_model_compare_states", "ddo_model_export_misp", "ddo_mdd_create", "ddo_mdd_destroy", "ddo_mdd_compile",
    "ddo_mdd_compile_batch", "ddo_mdd_is_exact", "ddo_mdd_best_value", "ddo_mdd_best_exact_value",
    "ddo_mdd_best_solution", "ddo_mdd_best_exact_solution", "ddo_mdd_drain_cutset", "ddo_mdd_drain_cutset_rows", "ddo_mdd_cutset_count",
    "ddo_mdd_last_counters", "ddo_mdd_combine_stats",
    "ddo_solver_create", "ddo_width_heuristic", "ddo_solver_destroy", "ddo_solver_maximize", "ddo_solver_best_value",
    "ddo_solver_best_solution", "ddo_solver_best_lower_bound", "ddo_solver_best_upper_bound", "ddo_solver_set_primal",
    "ddo_solver_gap", "ddo_solver_explored", "ddo_solver_counters", "ddo_solver_step", "ddo_solver_epoch", "ddo_solver_flush",
    "ddo_solver_import_lower_bound", "ddo_solver_fringe_len", "ddo_solver_fringe_best_ub", "ddo_solver_device_time", "ddo_solver_tier_count", "ddo_solver_tier_stats",
    "ddo_solver_bench_freeze", "ddo_solver_bench_step", "ddo_solver_bench_frozen",
    "ddo_solver_export_subproblems", "ddo_solver_import_subproblems",
    "ddo_dominance_create", "ddo_dominance_destroy", "ddo_dominance_clear", "ddo_cache_create", "ddo_cache_destroy", "ddo_cache_clear", "ddo_cache_stats", "ddo_cache_get_threshold", "ddo_cache_update_threshold",
]

DDO_OK, DDO_CUTOFF = 0, 2
LAST_EXACT_LAYER, FRONTIER = 1, 2
MDD_CACHING = 0x10
MDD_POOLED = 0x20
DDO_HANDED_UP = 3
MDD_ENGINES = {"auto": 0, "full": 0x400, "dense": 0x100, "tier0": 0x200, "tier1": 0x300}   # DDO_MDD_ENGINE_* (include/ddo_hip.h)


class _HandedUp:
    """ddo_mdd_compile answered DDO_HANDED_UP: the decision diagram does not fit the capacity tier the mdd is bound to."""

    def __repr__(self):
        return "HANDED_UP"


HANDED_UP = _HandedUp()


class DdoError(RuntimeError):
    pass


class CompilationType:  # mdd.rs:41-48
    Exact, Relaxed, Restricted = 0, 1, 2


class _Decision(C.Structure):
    _fields_ = [("variable", C.c_int64), ("value", C.c_int64)]


class _SubProblem(C.Structure):
    _fields_ = [("state", C.POINTER(C.c_uint64)), ("state_words", C.c_size_t), ("value", C.c_int64), ("ub", C.c_int64),
                ("depth", C.c_size_t), ("path", C.POINTER(_Decision)), ("path_len", C.c_size_t)]


class _CutsetRows(C.Structure):
    _fields_ = [("count", C.c_size_t), ("state_words", C.c_size_t), ("path_stride", C.c_size_t), ("states", C.POINTER(C.c_uint64)),
                ("values", C.POINTER(C.c_int64)), ("ubs", C.POINTER(C.c_int64)), ("depths", C.POINTER(C.c_size_t)),
                ("path_lens", C.POINTER(C.c_size_t)), ("paths", C.POINTER(_Decision))]


class _CompileInput(C.Structure):
    _fields_ = [("comp_type", C.c_int), ("max_width", C.c_size_t), ("best_lb", C.c_int64), ("residual", _SubProblem),
                ("cutoff", C.POINTER(C.c_int)), ("cache", C.c_void_p), ("dominance", C.c_void_p)]


class _Completion(C.Structure):
    _fields_ = [("is_exact", C.c_int), ("has_best_value", C.c_int), ("best_value", C.c_int64)]


class _Counters(C.Structure):
    _fields_ = [("nodes_expanded", C.c_uint64), ("arcs", C.c_uint64), ("layers", C.c_uint64), ("compiles", C.c_uint64)]


class _TierStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("launches", C.c_uint64), ("subproblems", C.c_uint64), ("retried", C.c_uint64),
                ("nodes_expanded", C.c_uint64), ("lds_bytes", C.c_uint64), ("layer_capacity", C.c_int32), ("threads", C.c_int32),
                ("slots", C.c_int32), ("dense", C.c_int32)]


class _SolverConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("width_policy", C.c_int), ("width", C.c_size_t), ("nb_concurrent", C.c_int),
                ("time_budget_s", C.c_double), ("rank", C.c_int), ("world_size", C.c_int), ("fringe", C.c_int),
                ("sequential", C.c_int), ("cutset_type", C.c_int), ("cache_entries", C.c_size_t), ("dominance_entries", C.c_size_t),
                ("width_times", C.c_size_t), ("width_div_by", C.c_size_t), ("pooled", C.c_int)]


_CUTSET_CB = C.CFUNCTYPE(None, C.POINTER(_SubProblem), C.c_void_p)

_lib = None


def library_path():
    # DDO_HIP_LIBRARY: another build of the same sources (A/B runs, diagnosis builds: make BUILD=../_build_x EXTRA=-D...)
    return os.environ.get("DDO_HIP_LIBRARY") or os.path.join(_HERE, "_build", "libddo_hip.so")


def lib():
    """Loads the HIP engine.  Fails loudly when it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise DdoError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); ddo_amd has no CPU fallback")
    L = C.CDLL(path)
    L.ddo_last_error.restype = C.c_char_p
    L.ddo_model_create_misp.restype = C.c_void_p
    L.ddo_model_create_misp.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.ddo_model_read_misp.restype = C.c_void_p
    L.ddo_model_read_misp.argtypes = [C.c_char_p]
    L.ddo_model_create_knapsack.restype = C.c_void_p
    L.ddo_model_create_knapsack.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    L.ddo_model_read_knapsack.restype = C.c_void_p
    L.ddo_model_read_knapsack.argtypes = [C.c_char_p]
    L.ddo_model_create_mcp.restype = C.c_void_p
    L.ddo_model_create_mcp.argtypes = [C.c_int, C.c_void_p]
    L.ddo_model_read_mcp.restype = C.c_void_p
    L.ddo_model_read_mcp.argtypes = [C.c_char_p]
    L.ddo_model_create_max2sat.restype = C.c_void_p
    L.ddo_model_create_max2sat.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddo_model_read_max2sat.restype = C.c_void_p
    L.ddo_model_read_max2sat.argtypes = [C.c_char_p]
    L.ddo_model_create_tsptw.restype = C.c_void_p
    L.ddo_model_create_tsptw.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddo_model_read_tsptw.restype = C.c_void_p
    L.ddo_model_read_tsptw.argtypes = [C.c_char_p]
    L.ddo_model_destroy.argtypes = [C.c_void_p]
    L.ddo_model_nb_variables.argtypes = [C.c_void_p]
    L.ddo_model_state_words.argtypes = [C.c_void_p]
    L.ddo_model_initial_state.argtypes = [C.c_void_p, C.c_void_p]
    L.ddo_model_initial_value.restype = C.c_int64
    L.ddo_model_initial_value.argtypes = [C.c_void_p]
    L.ddo_model_compare_states.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddo_model_export_misp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddo_dominance_create.restype = C.c_void_p
    L.ddo_dominance_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.ddo_dominance_destroy.argtypes = [C.c_void_p]
    L.ddo_dominance_clear.argtypes = [C.c_void_p]
    L.ddo_cache_create.restype = C.c_void_p
    L.ddo_cache_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.ddo_cache_destroy.argtypes = [C.c_void_p]
    L.ddo_cache_clear.argtypes = [C.c_void_p]
    L.ddo_cache_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ddo_cache_get_threshold.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.ddo_cache_update_threshold.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int]
    L.ddo_mdd_create.restype = C.c_void_p
    L.ddo_mdd_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t]
    L.ddo_mdd_destroy.argtypes = [C.c_void_p]
    L.ddo_mdd_compile.argtypes = [C.c_void_p, C.POINTER(_CompileInput), C.POINTER(_Completion)]
    L.ddo_mdd_compile_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(_CompileInput), C.POINTER(_Completion),
                                        C.POINTER(C.c_int), C.c_size_t]
    L.ddo_mdd_is_exact.argtypes = [C.c_void_p]
    L.ddo_mdd_best_value.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.ddo_mdd_best_exact_value.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.ddo_mdd_best_solution.argtypes = [C.c_void_p, C.POINTER(_Decision), C.POINTER(C.c_size_t)]
    L.ddo_mdd_best_exact_solution.argtypes = [C.c_void_p, C.POINTER(_Decision), C.POINTER(C.c_size_t)]
    L.ddo_mdd_drain_cutset.argtypes = [C.c_void_p, _CUTSET_CB, C.c_void_p]
    if hasattr(L, "ddo_mdd_drain_cutset_rows"):   # (DDO_HIP_LIBRARY may name an older diagnosis build: tools/diag)
        L.ddo_mdd_drain_cutset_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ddo_mdd_cutset_count.restype = C.c_size_t
        L.ddo_mdd_cutset_count.argtypes = [C.c_void_p]
    L.ddo_mdd_last_counters.argtypes = [C.c_void_p, C.POINTER(_Counters)]
    L.ddo_mdd_combine_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.ddo_solver_create.restype = C.c_void_p
    L.ddo_solver_create.argtypes = [C.c_void_p, C.POINTER(_SolverConfig)]
    L.ddo_width_heuristic.restype = C.c_size_t
    L.ddo_width_heuristic.argtypes = [C.POINTER(_SolverConfig), C.c_size_t, C.c_size_t]
    L.ddo_solver_destroy.argtypes = [C.c_void_p]
    L.ddo_solver_epoch.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.c_double]
    L.ddo_solver_maximize.argtypes = [C.c_void_p, C.POINTER(_Completion)]
    L.ddo_solver_best_value.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.ddo_solver_best_solution.argtypes = [C.c_void_p, C.POINTER(_Decision), C.POINTER(C.c_size_t)]
    L.ddo_solver_best_lower_bound.restype = C.c_int64
    L.ddo_solver_best_lower_bound.argtypes = [C.c_void_p]
    L.ddo_solver_best_upper_bound.restype = C.c_int64
    L.ddo_solver_best_upper_bound.argtypes = [C.c_void_p]
    L.ddo_solver_set_primal.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_Decision), C.c_size_t]
    L.ddo_solver_gap.restype = C.c_double
    L.ddo_solver_gap.argtypes = [C.c_void_p]
    L.ddo_solver_explored.restype = C.c_uint64
    L.ddo_solver_explored.argtypes = [C.c_void_p]
    L.ddo_solver_counters.argtypes = [C.c_void_p, C.POINTER(_Counters)]
    L.ddo_solver_step.argtypes = [C.c_void_p]
    L.ddo_solver_flush.argtypes = [C.c_void_p]
    L.ddo_solver_bench_freeze.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ddo_solver_bench_frozen.restype = C.c_uint64
    L.ddo_solver_bench_frozen.argtypes = [C.c_void_p]
    L.ddo_solver_bench_step.argtypes = [C.c_void_p]
    L.ddo_solver_export_subproblems.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ddo_solver_import_subproblems.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
    L.ddo_solver_import_lower_bound.argtypes = [C.c_void_p, C.c_int64]
    L.ddo_solver_fringe_len.restype = C.c_uint64
    L.ddo_solver_fringe_len.argtypes = [C.c_void_p]
    L.ddo_solver_fringe_best_ub.restype = C.c_int64
    L.ddo_solver_fringe_best_ub.argtypes = [C.c_void_p]
    L.ddo_solver_device_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.ddo_solver_tier_count.argtypes = [C.c_void_p]
    L.ddo_solver_tier_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(_TierStats)]
    _lib = L
    return L


def _err():
    return lib().ddo_last_error().decode(errors="replace")


def device_count():
    return lib().ddo_device_count()


# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Decision:  # common.rs:58-61
    variable: int
    value: int


@dataclass
class SubProblem:  # common.rs:75-87
    state: np.ndarray
    value: int = 0
    path: List[Decision] = field(default_factory=list)
    ub: int = (1 << 63) - 1
    depth: int = 0


@dataclass
class Completion:  # common.rs:115-121
    is_exact: bool
    best_value: Optional[int]


class FixedWidth:  # width.rs:166-171
    def __init__(self, w):
        self.w = int(w)


class Times:  # width.rs:636-642: max(1, k * inner)
    def __init__(self, k, inner):
        self.k, self.inner = int(k), inner


class DivBy:  # width.rs:875-881: max(1, inner / k)
    def __init__(self, k, inner):
        if int(k) < 1:
            raise ZeroDivisionError("DivBy(0, ..) divides by zero (width.rs:1073 panics)")
        self.k, self.inner = int(k), inner


def _width_config(cfg, width):
    """Fills width_policy / width / width_times / width_div_by of a _SolverConfig from a WidthHeuristic object:
    FixedWidth, NbUnassignedWidth, TsptwWidth, optionally wrapped as Times(k, inner), DivBy(k, inner) or DivBy(k, Times(j, inner))."""
    cfg.width_times = cfg.width_div_by = 0
    if isinstance(width, DivBy):
        cfg.width_div_by, width = width.k, width.inner
    if isinstance(width, Times):
        if width.k == 0:   # Times(0, inner) is the constant 1
            width = FixedWidth(1)
        else:
            cfg.width_times, width = width.k, width.inner
    if isinstance(width, FixedWidth):
        cfg.width_policy, cfg.width = 0, width.w
    elif isinstance(width, NbUnassignedWidth):
        cfg.width_policy, cfg.width = 1, 0
    elif isinstance(width, TsptwWidth):
        cfg.width_policy, cfg.width = 2, width.factor
    else:
        raise TypeError("width must be FixedWidth, NbUnassignedWidth or TsptwWidth, optionally inside Times(k, ..) / DivBy(k, ..) / DivBy(k, Times(j, ..))")
    return cfg


def width_heuristic(width, nb_vars, depth):
    """WidthHeuristic::max_width as the solver host evaluates it (ddo_width_heuristic; no device needed)."""
    cfg = _width_config(_SolverConfig(), width)
    return int(lib().ddo_width_heuristic(C.byref(cfg), int(nb_vars), int(depth)))


class NbUnassignedWidth:  # width.rs:397-402
    def __init__(self, nb_vars):
        self.nb_vars = int(nb_vars)


class NoCutoff:  # cutoff.rs:160-163
    seconds = 0.0


class TimeBudget:  # cutoff.rs:302-323
    def __init__(self, seconds):
        self.seconds = float(seconds)


class Misp:
    """MISP model == `Misp` + `MispRelax` + `MispRanking` (examples/misp/main.rs:37-209)."""

    def __init__(self, handle):
        if not handle:
            raise DdoError("could not create the MISP model: " + _err())
        self._h = handle
        L = lib()
        self.n = L.ddo_model_nb_variables(handle)
        self.ws = L.ddo_model_state_words(handle)

    @classmethod
    def read_instance(cls, path):  # main.rs:258
        return cls(lib().ddo_model_read_misp(os.fspath(path).encode()))

    @classmethod
    def from_rows(cls, n, rows, weights):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        weights = np.ascontiguousarray(weights, dtype=np.int64)
        return cls(lib().ddo_model_create_misp(n, rows.ctypes.data_as(C.c_void_p), weights.ctypes.data_as(C.c_void_p)))

    def __del__(self):
        try:
            if self._h:
                lib().ddo_model_destroy(self._h)
        except Exception:
            pass

    def nb_variables(self):
        return self.n

    def initial_state(self):
        s = np.zeros(self.ws, dtype=np.uint64)
        lib().ddo_model_initial_state(self._h, s.ctypes.data_as(C.c_void_p))
        return s

    def initial_value(self):
        return lib().ddo_model_initial_value(self._h)

    def compare(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        return lib().ddo_model_compare_states(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))

    def export(self):
        rows = np.zeros(self.n * self.ws, dtype=np.uint64)
        w = np.zeros(self.n, dtype=np.int64)
        lib().ddo_model_export_misp(self._h, rows.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
        return rows, w

    def root(self):
        return SubProblem(state=self.initial_state(), value=self.initial_value(), path=[], depth=0)


def _fill_input(model, comp_type, max_width, residual, best_lb, keep, cutoff=None, cache=None, dominance=None):
    ci = _CompileInput()
    ci.comp_type = comp_type
    ci.max_width = int(max_width)
    ci.best_lb = int(max(min(best_lb, (1 << 63) - 1), -(1 << 63)))
    st = np.ascontiguousarray(residual.state, dtype=np.uint64)
    path = (_Decision * max(1, len(residual.path)))()
    for i, d in enumerate(residual.path):
        path[i].variable, path[i].value = d.variable, d.value
    keep.extend([st, path])
    ci.residual.state = st.ctypes.data_as(C.POINTER(C.c_uint64))
    ci.residual.state_words = model.ws
    ci.residual.value = int(residual.value)
    ci.residual.ub = int(min(residual.ub, (1 << 63) - 1))
    ci.residual.depth = int(residual.depth)
    ci.residual.path = path
    ci.residual.path_len = len(residual.path)
    ci.cutoff = cutoff   # None, or a ctypes c_int polled like Cutoff::must_stop (clean.rs:352)
    ci.cache = cache._h if cache is not None else None
    ci.dominance = dominance._h if dominance is not None else None
    return ci



class Knapsack(Misp):
    """Knapsack model == `Knapsack` + `KPRelax` + `KPRanking` (examples/knapsack/main.rs:53-194); the state is two
    words, `KnapsackState { capacity, depth }`.  Shares the generic accessors of the model handle with `Misp`."""

    def __init__(self, handle):
        if not handle:
            raise DdoError("could not create the knapsack model: " + _err())
        self._h = handle
        L = lib()
        self.n = L.ddo_model_nb_variables(handle)
        self.ws = L.ddo_model_state_words(handle)

    @classmethod
    def read_instance(cls, path):  # main.rs:267
        return cls(lib().ddo_model_read_knapsack(os.fspath(path).encode()))

    @classmethod
    def from_items(cls, capacity, profit, weight):
        profit = np.ascontiguousarray(profit, dtype=np.int64)
        weight = np.ascontiguousarray(weight, dtype=np.int64)
        return cls(lib().ddo_model_create_knapsack(len(profit), int(capacity), profit.ctypes.data_as(C.c_void_p),
                                                   weight.ctypes.data_as(C.c_void_p)))


class Mcp(Misp):
    """Maximum-cut model == `Mcp` + `McpRelax` + `McpRanking` (examples/mcp); the state is n signed benefits (two per
    word) followed by a depth word; decisions are +1 (side S) / -1 (side T)."""

    def __init__(self, handle):
        if not handle:
            raise DdoError("could not create the max-cut model: " + _err())
        self._h = handle
        L = lib()
        self.n = L.ddo_model_nb_variables(handle)
        self.ws = L.ddo_model_state_words(handle)

    @classmethod
    def read_instance(cls, path):  # graph.rs:48
        return cls(lib().ddo_model_read_mcp(os.fspath(path).encode()))

    @classmethod
    def from_matrix(cls, adj):
        adj = np.ascontiguousarray(adj, dtype=np.int64)
        assert adj.ndim == 2 and adj.shape[0] == adj.shape[1]
        return cls(lib().ddo_model_create_mcp(adj.shape[0], adj.ctypes.data_as(C.c_void_p)))


class Max2Sat(Misp):
    """Weighted MAX2SAT model == `Max2Sat` + `Max2SatRelax` + `Max2SatRanking` (examples/max2sat); state and decisions as
    for `Mcp` (+1 = true, -1 = false)."""

    def __init__(self, handle):
        if not handle:
            raise DdoError("could not create the MAX2SAT model: " + _err())
        self._h = handle
        L = lib()
        self.n = L.ddo_model_nb_variables(handle)
        self.ws = L.ddo_model_state_words(handle)

    @classmethod
    def read_instance(cls, path):  # data.rs:67
        return cls(lib().ddo_model_read_max2sat(os.fspath(path).encode()))

    @classmethod
    def from_clauses(cls, n, clauses):
        """clauses: iterable of (lit_a, lit_b, weight)"""
        cl = list(clauses)
        a = np.ascontiguousarray([c[0] for c in cl], dtype=np.int64)
        b = np.ascontiguousarray([c[1] for c in cl], dtype=np.int64)
        w = np.ascontiguousarray([c[2] for c in cl], dtype=np.int64)
        return cls(lib().ddo_model_create_max2sat(n, len(cl), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                  w.ctypes.data_as(C.c_void_p)))

class SimpleCache:
    """`SimpleCache` (cache/simple.rs:36-73) in device memory: one table of (depth, state) -> Threshold shared by every
    compile that names it."""

    def __init__(self, model, capacity=1 << 20, device=0):
        self.model = model
        self._h = lib().ddo_cache_create(model._h, device, int(capacity))
        if not self._h:
            raise DdoError("ddo_cache_create failed: " + _err())

    def __del__(self):
        try:
            if self._h:
                lib().ddo_cache_destroy(self._h)
        except Exception:
            pass

    def clear(self):  # cache.rs:56
        lib().ddo_cache_clear(self._h)

    def stats(self):
        u, d = C.c_uint64(), C.c_uint64()
        lib().ddo_cache_stats(self._h, C.byref(u), C.byref(d))
        return {"used": u.value, "dropped": d.value}

    def get_threshold(self, state, depth):  # cache.rs:44 -> (value, explored) or None
        st = np.ascontiguousarray(state, dtype=np.uint64)
        v, e = C.c_int64(), C.c_int()
        rc = lib().ddo_cache_get_threshold(self._h, st.ctypes.data_as(C.c_void_p), int(depth), C.byref(v), C.byref(e))
        if rc < 0:
            raise DdoError(f"ddo_cache_get_threshold rc={rc}: {_err()}")
        return (v.value, bool(e.value)) if rc == 1 else None

    def update_threshold(self, state, depth, value, explored):  # cache.rs:47
        st = np.ascontiguousarray(state, dtype=np.uint64)
        rc = lib().ddo_cache_update_threshold(self._h, st.ctypes.data_as(C.c_void_p), int(depth), int(value), 1 if explored else 0)
        if rc < 0:
            raise DdoError(f"ddo_cache_update_threshold rc={rc}: {_err()}")


class Tsptw(Misp):
    """TSPTW model == `Tsptw` + `TsptwRelax` + `TsptwRanking` (examples/tsptw), up to 256 nodes; 5 / 8 / 14-word states (include/ddo_hip.h); variable k is the
    k-th move of the tour, its value the node visited; values are MINUS the elapsed time in 1/10000 units."""

    def __init__(self, handle):
        if not handle:
            raise DdoError("could not create the TSPTW model: " + _err())
        self._h = handle
        L = lib()
        self.n = L.ddo_model_nb_variables(handle)
        self.ws = L.ddo_model_state_words(handle)

    @classmethod
    def read_instance(cls, path):  # instance.rs:52-109
        return cls(lib().ddo_model_read_tsptw(os.fspath(path).encode()))

    @classmethod
    def from_arrays(cls, distances, earliest, latest):
        d = np.ascontiguousarray(distances, dtype=np.int64)
        e = np.ascontiguousarray(earliest, dtype=np.int64)
        l = np.ascontiguousarray(latest, dtype=np.int64)
        assert d.ndim == 2 and d.shape[0] == d.shape[1] == len(e) == len(l)
        return cls(lib().ddo_model_create_tsptw(d.shape[0], d.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p)))


class TsptwWidth:  # examples/tsptw/heuristics.rs:38-52: nb_vars * (depth + 1) * factor
    def __init__(self, factor=1):
        self.factor = int(factor)


class SimpleDominanceChecker:
    """`SimpleDominanceChecker::new(KPDominance, nb_variables)` (dominance/simple.rs:37-117, examples/knapsack/main.rs:198-218,
    325) in device memory: knapsack models only (the relation is keyed by the depth, one coordinate + the value)."""

    def __init__(self, model, capacity_per_depth=1 << 14, device=0):
        self.model = model
        self._h = lib().ddo_dominance_create(model._h, device, int(capacity_per_depth))
        if not self._h:
            raise DdoError("ddo_dominance_create failed: " + _err())

    def __del__(self):
        try:
            if self._h:
                lib().ddo_dominance_destroy(self._h)
        except Exception:
            pass

    def clear(self):
        lib().ddo_dominance_clear(self._h)


class Mdd:
    """`impl DecisionDiagram for Mdd<T, CUTSET_TYPE>` (mdd.rs:75-114) on the device: LAST_EXACT_LAYER or FRONTIER cut-set;
    caching=True lets compile() take a SimpleCache."""

    def __init__(self, model, max_width, device=0, cutset_type=LAST_EXACT_LAYER, caching=False, engine="auto"):
        self.model = model
        self._h = lib().ddo_mdd_create(model._h, device, cutset_type | (MDD_CACHING if caching else 0) | MDD_ENGINES[engine], int(max_width))
        if not self._h:
            raise DdoError("ddo_mdd_create failed: " + _err())

    def __del__(self):
        try:
            if self._h:
                lib().ddo_mdd_destroy(self._h)
        except Exception:
            pass

    def compile(self, comp_type, max_width, residual, best_lb, cutoff=None, cache=None, dominance=None):
        keep = []
        ci = _fill_input(self.model, comp_type, max_width, residual, best_lb, keep,
                         C.pointer(cutoff) if cutoff is not None else None, cache, dominance)   # cutoff: ctypes.c_int
        out = _Completion()
        rc = lib().ddo_mdd_compile(self._h, C.byref(ci), C.byref(out))
        if rc == DDO_CUTOFF:
            return None  # Err(Reason::CutoffOccurred)
        if rc == DDO_HANDED_UP:
            return HANDED_UP
        if rc != DDO_OK:
            raise DdoError(f"ddo_mdd_compile rc={rc}: {_err()}")
        return Completion(bool(out.is_exact), out.best_value if out.has_best_value else None)

    @staticmethod
    def compile_batch(mdds, comp_types, max_widths, residuals, best_lbs, cache=None, cutoffs=None):
        """`cutoffs`: one ctypes c_int (or None) per compile, polled like Cutoff::must_stop; a compile stopped by ITS flag comes
        back as None, the others run to the end whatever their neighbours' flags do."""
        n = len(mdds)
        keep = []
        cis = (_CompileInput * n)()
        for i in range(n):
            cut = C.pointer(cutoffs[i]) if cutoffs is not None and cutoffs[i] is not None else None
            cis[i] = _fill_input(mdds[i].model, comp_types[i], max_widths[i], residuals[i], best_lbs[i], keep, cut, cache)
        hs = (C.c_void_p * n)(*[m._h for m in mdds])
        outs = (_Completion * n)()
        sts = (C.c_int * n)()
        rc = lib().ddo_mdd_compile_batch(hs, cis, outs, sts, n)
        if rc < 0:
            raise DdoError(f"ddo_mdd_compile_batch rc={rc}: {_err()} statuses={list(sts)}")
        return [HANDED_UP if sts[i] == DDO_HANDED_UP else None if sts[i] == DDO_CUTOFF else
                Completion(bool(o.is_exact), o.best_value if o.has_best_value else None) for i, o in enumerate(outs)]

    def is_exact(self):
        return bool(lib().ddo_mdd_is_exact(self._h))

    def best_value(self):
        v = C.c_int64()
        return v.value if lib().ddo_mdd_best_value(self._h, C.byref(v)) == 1 else None

    def best_exact_value(self):
        v = C.c_int64()
        return v.value if lib().ddo_mdd_best_exact_value(self._h, C.byref(v)) == 1 else None

    def _solution(self, fn):
        cap = 2 * self.model.n + 8
        buf = (_Decision * cap)()
        ln = C.c_size_t(cap)
        rc = fn(self._h, buf, C.byref(ln))
        if rc == 0:
            return None
        if rc != 1:
            raise DdoError(f"solution query rc={rc}: {_err()}")
        return [Decision(buf[i].variable, buf[i].value) for i in range(ln.value)]

    def best_solution(self):
        return self._solution(lib().ddo_mdd_best_solution)

    def best_exact_solution(self):
        return self._solution(lib().ddo_mdd_best_exact_solution)

    def drain_cutset(self, func=None):
        out = []
        ws = self.model.ws

        def cb(sp, _user):
            s = sp.contents
            state = np.array([s.state[k] for k in range(ws)], dtype=np.uint64)
            path = [Decision(s.path[i].variable, s.path[i].value) for i in range(s.path_len)]
            node = SubProblem(state=state, value=s.value, path=path, ub=s.ub, depth=s.depth)
            out.append(node)
            if func:
                func(node)

        rc = lib().ddo_mdd_drain_cutset(self._h, _CUTSET_CB(cb), None)
        if rc != DDO_OK:
            raise DdoError(f"ddo_mdd_drain_cutset rc={rc}: {_err()}")
        return out

    def cutset_count(self):
        return int(lib().ddo_mdd_cutset_count(self._h))

    def drain_cutset_rows(self, ub_above=-(1 << 63), residual_path=()):
        """ddo_mdd_drain_cutset_rows: the cut-set as flat rows (nodes with ub <= ub_above left out), rebuilt here into the
        SubProblems drain_cutset's callbacks deliver -- the residual's own path goes in front of the DD's part."""
        rows = _CutsetRows()
        rc = lib().ddo_mdd_drain_cutset_rows(self._h, C.c_int64(ub_above), C.byref(rows))
        if rc != DDO_OK:
            raise DdoError(f"ddo_mdd_drain_cutset_rows rc={rc}: {_err()}")
        out = []
        ws, st = rows.state_words, rows.path_stride
        head = list(residual_path)
        for i in range(rows.count):
            state = np.array([rows.states[i * ws + k] for k in range(ws)], dtype=np.uint64)
            path = head + [Decision(rows.paths[i * st + k].variable, rows.paths[i * st + k].value) for k in range(rows.path_lens[i])]
            out.append(SubProblem(state=state, value=rows.values[i], path=path, ub=rows.ubs[i], depth=rows.depths[i]))
        return out

    def combine_stats(self):
        """(device launches, compiles, kernel ms) that went through the combining layer of this mdd's engine (include/ddo_hip.h)"""
        a, b, k = C.c_uint64(), C.c_uint64(), C.c_double()
        lib().ddo_mdd_combine_stats(self._h, C.byref(a), C.byref(b), C.byref(k))
        return a.value, b.value, k.value

    def counters(self):
        c = _Counters()
        lib().ddo_mdd_last_counters(self._h, C.byref(c))
        return {"nodes_expanded": c.nodes_expanded, "arcs": c.arcs, "layers": c.layers, "compiles": c.compiles}


DefaultMDD = DefaultMDDLEL = Mdd  # mdd/mod.rs:42-49


def DefaultMDDFC(model, max_width, device=0, caching=False):  # mdd/mod.rs:46-49
    return Mdd(model, max_width, device=device, cutset_type=FRONTIER, caching=caching)


class ParallelSolver:
    """`ParallelSolver::custom(problem, relaxation, ranking, width, dominance, cutoff, fringe, nb_threads)`
    (parallel.rs:320-358); relaxation / ranking are carried by the model, dominance is the empty checker,
    the fringe is the NoDupFringe<MaxUB>.  `nb_threads` = sub-problems compiled concurrently on the GPU."""

    def __init__(self, problem, width, cutoff=None, nb_threads=256, device=0, rank=0, world_size=1, fringe="nodup",
                 sequential=False, cutset_type=LAST_EXACT_LAYER, cache_entries=0, dominance_entries=0, pooled=False):
        self.problem = problem
        cfg = _SolverConfig()
        cfg.device = device
        _width_config(cfg, width)
        cfg.nb_concurrent = int(nb_threads)
        cfg.time_budget_s = float(getattr(cutoff, "seconds", 0.0) or 0.0)
        cfg.rank, cfg.world_size = int(rank), int(world_size)
        cfg.fringe = {"nodup": 0, "lazy": 1}[fringe]  # NoDupFringe (exact ddo order) | lazy block SimpleFringe on device
        cfg.sequential = 1 if sequential else 0
        cfg.cutset_type = int(cutset_type)          # the `D` of ParallelSolver<State, D, C>: DefaultMDDLEL | DefaultMDDFC
        cfg.cache_entries = int(cache_entries)      # the `C`: 0 = EmptyCache, else SimpleCache with that many entries on the device
        cfg.dominance_entries = int(dominance_entries)   # 0 = EmptyDominanceChecker, else SimpleDominanceChecker (knapsack models)
        cfg.pooled = 1 if pooled else 0             # `D` = Pooled (mdd/pooled.rs): MISP, NoDupFringe, EmptyCache
        self._h = lib().ddo_solver_create(problem._h, C.byref(cfg))
        if not self._h:
            raise DdoError("ddo_solver_create failed: " + _err())

    custom = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def __del__(self):
        try:
            if self._h:
                lib().ddo_solver_destroy(self._h)
        except Exception:
            pass

    def maximize(self):
        out = _Completion()
        rc = lib().ddo_solver_maximize(self._h, C.byref(out))
        if rc < 0:
            raise DdoError(f"ddo_solver_maximize rc={rc}: {_err()}")
        return Completion(bool(out.is_exact), out.best_value if out.has_best_value else None)

    def step(self):
        rc = lib().ddo_solver_step(self._h)
        if rc < 0:
            raise DdoError(f"ddo_solver_step rc={rc}: {_err()}")
        return rc

    def epoch(self, prev, max_steps=64, min_ms=2.0):
        """one epoch of a sharded search (include/ddo_hip.h: ddo_solver_epoch): `prev` = the reduced 7-vector of the previous epoch or None;
        returns (rc of the last step, this rank's 7-vector for the next MAX all-reduce)"""
        vin = (C.c_int64 * 7)(*[int(x) for x in prev]) if prev is not None else None
        vout = (C.c_int64 * 7)()
        rc = lib().ddo_solver_epoch(self._h, vin, vout, int(max_steps), float(min_ms))
        if rc < 0:
            raise DdoError(f"ddo_solver_epoch rc={rc}: {_err()}")
        return rc, [int(x) for x in vout]

    def flush(self):
        rc = lib().ddo_solver_flush(self._h)
        if rc < 0:
            raise DdoError(f"ddo_solver_flush rc={rc}: {_err()}")
        return rc

    def bench_frozen(self):
        return lib().ddo_solver_bench_frozen(self._h)

    def bench_freeze(self, nbatches, stride=1):
        """Measurement support: freeze `nbatches` batches made of every stride-th node of the fringe (see include/ddo_hip.h)."""
        rc = lib().ddo_solver_bench_freeze(self._h, int(nbatches), int(stride))
        if rc < 0:
            raise DdoError(f"ddo_solver_bench_freeze rc={rc}: {_err()}")
        return rc

    def bench_step(self):
        rc = lib().ddo_solver_bench_step(self._h)
        if rc < 0:
            raise DdoError(f"ddo_solver_bench_step rc={rc}: {_err()}")
        return rc

    def export_subproblems(self, max_count):
        """Work hand-over (include/ddo_hip.h): pops up to max_count open sub-problems; returns a dict of numpy arrays
        {states [k, ws] u64, value, ub, depth [k] i64, path_off [k+1] u64, paths [total, 2] i64 (variable, value)}."""
        max_count = int(max_count)
        ws, n = self.problem.ws, self.problem.n
        states = np.zeros((max(1, max_count), ws), dtype=np.uint64)
        value = np.zeros(max(1, max_count), dtype=np.int64)
        ub = np.zeros(max(1, max_count), dtype=np.int64)
        depth = np.zeros(max(1, max_count), dtype=np.int64)
        path_off = np.zeros(max_count + 1, dtype=np.uint64)
        cap = max(1, max_count) * n + n
        paths = np.zeros((cap, 2), dtype=np.int64)
        cnt = C.c_size_t(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib().ddo_solver_export_subproblems(self._h, max_count, p(states), p(value), p(ub), p(depth), p(path_off), p(paths), cap,
                                                 C.byref(cnt))
        if rc < 0:
            raise DdoError(f"ddo_solver_export_subproblems rc={rc}: {_err()}")
        k = cnt.value
        tot = int(path_off[k]) if k else 0
        return {"states": states[:k].copy(), "value": value[:k].copy(), "ub": ub[:k].copy(), "depth": depth[:k].copy(),
                "path_off": path_off[:k + 1].copy(), "paths": paths[:tot].copy()}

    def import_subproblems(self, nodes):
        """Takes over sub-problems another solver of the same model exported (dict as returned by export_subproblems)."""
        k = len(nodes["value"])
        if k == 0:
            return
        arr = {key: np.ascontiguousarray(nodes[key], dtype=dt) for key, dt in
               (("states", np.uint64), ("value", np.int64), ("ub", np.int64), ("depth", np.int64), ("path_off", np.uint64), ("paths", np.int64))}
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib().ddo_solver_import_subproblems(self._h, k, p(arr["states"]), p(arr["value"]), p(arr["ub"]), p(arr["depth"]),
                                                 p(arr["path_off"]), p(arr["paths"]))
        if rc < 0:
            raise DdoError(f"ddo_solver_import_subproblems rc={rc}: {_err()}")

    def best_value(self):
        v = C.c_int64()
        return v.value if lib().ddo_solver_best_value(self._h, C.byref(v)) == 1 else None

    def best_solution(self):
        cap = 2 * self.problem.n + 8
        buf = (_Decision * cap)()
        ln = C.c_size_t(cap)
        rc = lib().ddo_solver_best_solution(self._h, buf, C.byref(ln))
        if rc != 1:
            return None
        return [Decision(buf[i].variable, buf[i].value) for i in range(ln.value)]

    def best_lower_bound(self):
        return lib().ddo_solver_best_lower_bound(self._h)

    def best_upper_bound(self):
        return lib().ddo_solver_best_upper_bound(self._h)

    def set_primal(self, value, solution):
        buf = (_Decision * max(1, len(solution)))()
        for i, d in enumerate(solution):
            buf[i].variable, buf[i].value = d.variable, d.value
        lib().ddo_solver_set_primal(self._h, int(value), buf, len(solution))

    def import_lower_bound(self, lb):
        lib().ddo_solver_import_lower_bound(self._h, int(lb))

    def gap(self):
        return lib().ddo_solver_gap(self._h)

    def explored(self):
        return lib().ddo_solver_explored(self._h)

    def fringe_len(self):
        return lib().ddo_solver_fringe_len(self._h)

    def fringe_best_ub(self):
        return lib().ddo_solver_fringe_best_ub(self._h)

    def counters(self):
        c = _Counters()
        lib().ddo_solver_counters(self._h, C.byref(c))
        return {"nodes_expanded": c.nodes_expanded, "arcs": c.arcs, "layers": c.layers, "compiles": c.compiles}

    def tier_stats(self):
        """per-tier accounting (`ddo_tier_stats`): capacity tiers first, then the dense tier, the full-width engine last"""
        out = []
        for t in range(lib().ddo_solver_tier_count(self._h)):
            ts = _TierStats()
            if lib().ddo_solver_tier_stats(self._h, t, C.byref(ts)) == 0:
                out.append({name: getattr(ts, name) for name, _ in _TierStats._fields_})
        return out

    def device_time(self):
        ms = C.c_double()
        n = C.c_uint64()
        lib().ddo_solver_device_time(self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value


DefaultSolver = ParallelSolver


def DefaultCachingSolver(problem, width, cutoff=None, nb_threads=256, device=0, cache_entries=1 << 22, cutset_type=FRONTIER, **kw):
    """`DefaultCachingSolver` (solver/mod.rs): ParallelSolver<State, DefaultMDDFC<State>, SimpleCache<State>>"""
    return ParallelSolver(problem, width, cutoff, nb_threads=nb_threads, device=device, cutset_type=cutset_type, cache_entries=cache_entries, **kw)


class SequentialSolver(ParallelSolver):
    """SequentialSolver (sequential.rs:202-527): one sub-problem at a time, NoDupFringe, and its `explored` bookkeeping."""

    def __init__(self, problem, width, cutoff=None, device=0, cutset_type=LAST_EXACT_LAYER, cache_entries=0, dominance_entries=0, pooled=False):
        super().__init__(problem, width, cutoff=cutoff, nb_threads=1, device=device, fringe="nodup", sequential=True,
                         cutset_type=cutset_type, cache_entries=cache_entries, dominance_entries=dominance_entries, pooled=pooled)
  # solver/mod.rs:28


# ---- the solver aliases of solver/mod.rs:29-47: (parallel | sequential) x (LEL | FC | Pooled) x (EmptyCache | SimpleCache)
def _alias(parallel, cutset_type, caching, doc):
    def make(problem, width, cutoff=None, nb_threads=256, device=0, cache_entries=1 << 22, **kw):
        ce = cache_entries if caching else 0
        if parallel:
            return ParallelSolver(problem, width, cutoff, nb_threads=nb_threads, device=device, cutset_type=cutset_type, cache_entries=ce, **kw)
        return SequentialSolver(problem, width, cutoff, device=device, cutset_type=cutset_type, cache_entries=ce, **kw)
    make.__doc__ = doc
    return make


ParNoCachingSolverLel = _alias(True, LAST_EXACT_LAYER, False, "ParallelSolver<State, DefaultMDDLEL<State>, EmptyCache<State>> (solver/mod.rs:32)")
ParNoCachingSolverFc = _alias(True, FRONTIER, False, "ParallelSolver<State, DefaultMDDFC<State>, EmptyCache<State>> (solver/mod.rs:33)")
ParCachingSolverLel = _alias(True, LAST_EXACT_LAYER, True, "ParallelSolver<State, DefaultMDDLEL<State>, SimpleCache<State>> (solver/mod.rs:36)")
ParCachingSolverFc = _alias(True, FRONTIER, True, "ParallelSolver<State, DefaultMDDFC<State>, SimpleCache<State>> (solver/mod.rs:37)")
SeqNoCachingSolverLel = _alias(False, LAST_EXACT_LAYER, False, "SequentialSolver<State, DefaultMDDLEL<State>, EmptyCache<State>> (solver/mod.rs:41)")
SeqNoCachingSolverFc = _alias(False, FRONTIER, False, "SequentialSolver<State, DefaultMDDFC<State>, EmptyCache<State>> (solver/mod.rs:42)")
SeqCachingSolverLel = _alias(False, LAST_EXACT_LAYER, True, "SequentialSolver<State, DefaultMDDLEL<State>, SimpleCache<State>> (solver/mod.rs:45)")
SeqCachingSolverFc = _alias(False, FRONTIER, True, "SequentialSolver<State, DefaultMDDFC<State>, SimpleCache<State>> (solver/mod.rs:46)")


def Pooled(model, max_width, device=0, caching=False):
    """`Pooled<State>` (mdd/pooled.rs:117-823), the long-arc decision diagram, on the device: an Mdd whose layers hold the pool nodes
    the branching variable impacts (MISP models: the reference's only model with Problem::is_impacted_by, misp/main.rs:145-147);
    frontier cut-set, depth of a sub-problem = the layer at which its node was expanded, one decision per expanded ancestor.
    caching=True lets compile() take a SimpleCache (pooled.rs:467-535, 662-680)."""
    return Mdd(model, max_width, device=device, cutset_type=FRONTIER | MDD_POOLED, caching=caching)


def _pooled_alias(parallel, caching, doc):
    def make(problem, width, cutoff=None, nb_threads=256, device=0, cache_entries=1 << 22, **kw):
        ce = cache_entries if caching else 0
        if parallel:
            return ParallelSolver(problem, width, cutoff, nb_threads=nb_threads, device=device, pooled=True, cache_entries=ce, **kw)
        return SequentialSolver(problem, width, cutoff, device=device, pooled=True, cache_entries=ce, **kw)
    make.__doc__ = doc
    return make


ParNoCachingSolverPooled = _pooled_alias(True, False, "ParallelSolver<State, Pooled<State>, EmptyCache<State>> (solver/mod.rs:34)")
SeqNoCachingSolverPooled = _pooled_alias(False, False, "SequentialSolver<State, Pooled<State>, EmptyCache<State>> (solver/mod.rs:43)")
ParCachingSolverPooled = _pooled_alias(True, True, "ParallelSolver<State, Pooled<State>, SimpleCache<State>> (solver/mod.rs:38)")
SeqCachingSolverPooled = _pooled_alias(False, True, "SequentialSolver<State, Pooled<State>, SimpleCache<State>> (solver/mod.rs:47)")
