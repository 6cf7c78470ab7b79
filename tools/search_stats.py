"""Whole search with the lazy fringe and its statistics (DDO_HIP_STATS=1 adds the per-phase / per-DD lines):
    gpurun -- python tools/search_stats.py <instance> <width> <sub-problems in flight> [time budget s]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver
name, w, conc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
model = ddo_amd.Misp.read_instance(f"data/misp/{name}.clq")
budget = float(sys.argv[4]) if len(sys.argv) > 4 else 60.0
s = ParallelSolver(model, FixedWidth(w), ddo_amd.TimeBudget(budget), nb_threads=conc, fringe="lazy")
t0 = time.perf_counter(); c = s.maximize(); dt = time.perf_counter() - t0
cn = s.counters()
print(name, w, conc, c.best_value, c.is_exact, "explored", s.explored(), cn, round(dt, 3), "device", s.device_time(), flush=True)
del s
