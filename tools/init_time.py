import os, sys, time
sys.path.insert(0, os.getcwd())
t0 = time.perf_counter()
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver
t1 = time.perf_counter()
print(f"import {t1 - t0:.2f} s")
for name, w, conc, fr in (("brock200_2", 1000, 256, "nodup"), ("brock200_2", 1000, 256, "lazy"), ("brock400_1", 10000, 2048, "lazy"), ("brock400_1", 10000, 2048, "lazy")):
    m = ddo_amd.Misp.read_instance(f"data/misp/{name}.clq")
    t0 = time.perf_counter()
    s = ParallelSolver(m, FixedWidth(w), nb_threads=conc, fringe=fr)
    t1 = time.perf_counter()
    s.step(); s.flush()
    t2 = time.perf_counter()
    del s
    t3 = time.perf_counter()
    print(f"{name} w={w} {fr}: create {t1 - t0:.3f} s, first step {t2 - t1:.3f} s, destroy {t3 - t2:.3f} s", flush=True)
