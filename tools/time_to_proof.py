"""Secondary metric: wall time to the proved optimum, device engine vs the CPU oracle on the
same box, for the parity configurations C1 (knapsack), C2 (MISP brock200_2) and C3 (MAX2SAT frb10-6-1)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver
from tests.oracle_binding import Oracle

o = Oracle("oracle/_build/liboracle.so")


def gpu(model, width, conc, fringe):
    s = ParallelSolver(model, FixedWidth(width), nb_threads=conc, fringe=fringe)
    t0 = time.perf_counter()
    c = s.maximize()
    return c.best_value, s.explored(), round(time.perf_counter() - t0, 3)


# C2
path = "data/misp/brock200_2.clq"
model = ddo_amd.Misp.read_instance(path)
print("C2 gpu lazy x256 ", gpu(model, 1000, 256, "lazy"))
print("C2 gpu nodup x256", gpu(model, 1000, 256, "nodup"))
inst = o.misp(path)
for th in (1, 32):
    r = inst.solve(1000, th, 0)
    print("C2 cpu threads", th, (r["best_value"], r["explored"], round(r["wall_s"], 3)))
# C3
path = "data/max2sat/frb10-6-1.wcnf"
model = ddo_amd.Max2Sat.read_instance(path)
print("C3 gpu nodup x64 ", gpu(model, 5000, 64, "nodup"))
print("C3 gpu nodup x256", gpu(model, 5000, 256, "nodup"))
for th in (1, 32):
    v, info = o.max2sat_file(path, 5000, th)
    print("C3 cpu threads", th, (v, info["explored"], round(info["wall_s"], 3)))
# C1
sys.path.insert(0, "tests")
from test_gpu_knapsack import lcg_instance  # noqa: E402

cap, profit, weight = lcg_instance()
model = ddo_amd.Knapsack.from_items(cap, profit, weight)
print("C1 gpu nodup x1  ", gpu(model, 100, 1, "nodup"))
print("C1 gpu nodup x64 ", gpu(model, 100, 64, "nodup"))
v, info = o.knapsack(profit, weight, cap, 100, 0)
print("C1 cpu sequential", (v, info["explored"], round(info["wall_s"], 3)))
