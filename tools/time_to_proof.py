import os, sys, time
sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver
from tests.oracle_binding import Oracle
path = "data/misp/brock200_2.clq"
model = ddo_amd.Misp.read_instance(path)
for fr, conc in (("lazy", 1024), ("lazy", 256), ("nodup", 256)):
    s = ParallelSolver(model, FixedWidth(1000), nb_threads=conc, fringe=fr)
    t0 = time.perf_counter(); c = s.maximize(); t1 = time.perf_counter()
    print("gpu", fr, conc, c, "explored", s.explored(), "s", round(t1 - t0, 3), flush=True)
o = Oracle("oracle/_build/liboracle.so")
inst = o.misp(path)
for th in (1, 32):
    r = inst.solve(1000, th, 0)
    print("cpu threads", th, r["best_value"], "explored", r["explored"], "s", round(r["wall_s"], 3), flush=True)
