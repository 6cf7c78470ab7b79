import sys, time, os
sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver
name, w, conc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
model = ddo_amd.Misp.read_instance(f"data/misp/{name}.clq")
s = ParallelSolver(model, FixedWidth(w), ddo_amd.TimeBudget(60), nb_threads=conc, fringe="lazy")
t0 = time.perf_counter(); c = s.maximize(); dt = time.perf_counter() - t0
cn = s.counters()
print(name, w, conc, c.best_value, c.is_exact, "explored", s.explored(), cn, round(dt, 3), "device", s.device_time(), flush=True)
del s
