"""Runs the lazy proof search of brock400_1 / W = 10 000 for a time budget (tools: kernel traces of the live search)."""
import sys
import time

sys.path.insert(0, ".")
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver, TimeBudget

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
conc = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
model = ddo_amd.Misp.read_instance("data/misp/brock400_1.clq")
s = ParallelSolver(model, FixedWidth(10000), TimeBudget(budget), nb_threads=conc, fringe="lazy")
t0 = time.perf_counter()
c = s.maximize()
dt = time.perf_counter() - t0
k, l = s.device_time()
print("wall %.2f kernel %.2f launches %d explored %d nodes %d" % (dt, k / 1e3, l, s.explored(), s.counters()["nodes_expanded"]))
for t in s.tier_stats():
    print(t)
