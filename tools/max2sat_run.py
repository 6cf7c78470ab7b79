import sys, time
sys.path.insert(0, ".")
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver, TimeBudget
m = ddo_amd.Max2Sat.read_instance("data/max2sat/frb15-9-1.wcnf")
s = ParallelSolver(m, FixedWidth(5000), TimeBudget(float(sys.argv[1])), nb_threads=256, fringe="nodup")
t0 = time.perf_counter(); c = s.maximize(); dt = time.perf_counter() - t0
k, l = s.device_time(); cnt = s.counters()
print("wall %.2f kernel %.2f launches %d explored %d nodes %d -> %.3g nodes/s" % (dt, k / 1e3, l, s.explored(), cnt["nodes_expanded"], cnt["nodes_expanded"] / dt))
