"""ddo_amd -- MI355X-native MDD compilation engine behind ddo's API (MISP hot path).

Python here is plumbing only: a ctypes view of the C ABI in ``include/ddo_hip.h``
(``ddo_amd/_build/libddo_hip.so``, built by ``__graft_entry__.build()``) with the
names of the reference's own types, so tests read like the reference's tests
(/root/reference/ddo/examples/misp/tests.rs).  There is no CPU fallback: importing
works without a GPU (symbol checks), creating an ``Mdd``/solver without one raises.
"""
from .binding import (  # noqa: F401
    CompilationType, Completion, Decision, DdoError, FixedWidth, LAST_EXACT_LAYER, FRONTIER, Mdd, DefaultMDD,
    DefaultMDDLEL, DefaultMDDFC, DefaultCachingSolver, SimpleCache, SimpleDominanceChecker, Tsptw, TsptwWidth, Misp, Knapsack, Mcp, Max2Sat, NbUnassignedWidth, NoCutoff, ParallelSolver, SequentialSolver, DefaultSolver, SubProblem, TimeBudget, lib,
    library_path, device_count, ABI_SYMBOLS, HANDED_UP, MDD_ENGINES, Times, DivBy, width_heuristic,
    ParNoCachingSolverLel, ParNoCachingSolverFc, ParCachingSolverLel, ParCachingSolverFc, SeqNoCachingSolverLel, SeqNoCachingSolverFc,
    SeqCachingSolverLel, SeqCachingSolverFc, Pooled, ParNoCachingSolverPooled, ParCachingSolverPooled, SeqNoCachingSolverPooled,
    SeqCachingSolverPooled)
