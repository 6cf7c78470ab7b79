"""Diagnosis: what a SimpleCache that fills up costs a pooled search: ParCachingSolverPooled on brock200_4 under
NbUnassignedWidth without a cache, with the default 4 M entries, and with a table that never fills -- proved?, explored, compiles, wall and
kernel seconds, launches.   gpurun -- python tools/diag/pooled_cache_probe.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import NbUnassignedWidth, ParallelSolver
model = ddo_amd.Misp.read_instance("data/misp/brock200_4.clq")
for ce in (0, 1 << 22, 1 << 27):
    for rep in range(2):
        s = ParallelSolver(model, NbUnassignedWidth(model.n), ddo_amd.TimeBudget(40.0), nb_threads=256, fringe="nodup", pooled=True, cache_entries=ce)
        k0, l0 = s.device_time(); t0 = time.perf_counter(); c = s.maximize(); dt = time.perf_counter() - t0; k1, l1 = s.device_time()
        cnt = s.counters()
    print("cache_entries", ce, "proved", c.is_exact, "explored", s.explored(), "compiles", cnt["compiles"], "wall %.3f kernel %.3f launches %d nodes %d" % (dt, (k1 - k0) / 1e3, l1 - l0, cnt["nodes_expanded"]), flush=True)
