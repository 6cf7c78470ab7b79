"""Lazy block fringe (SimpleFringe semantics, device-resident) vs NoDupFringe (host) on whole searches: explored
sub-problems, nodes, wall time, for several numbers of sub-problems in flight.
    gpurun -- python tools/fringe_compare.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver

CASES = (("brock200_2", 1000), ("brock200_4", 1000), ("keller4", 100), ("p_hat300-1", 100), ("brock200_1", 2000), ("brock200_1", 10000))
CONC = (256,) if "quick" in sys.argv else (256, 1024, 4096)
for name, w in CASES:
    model = ddo_amd.Misp.read_instance(f"data/misp/{name}.clq")
    for conc in CONC:
        for fr in ("lazy", "nodup"):
            s = ParallelSolver(model, FixedWidth(w), ddo_amd.TimeBudget(60), nb_threads=conc, fringe=fr)
            t0 = time.perf_counter()
            c = s.maximize()
            dt = time.perf_counter() - t0
            nodes = s.counters()["nodes_expanded"]
            print(f"{name} w={w} x{conc} {fr}: value {c.best_value} exact {c.is_exact} explored {s.explored()} nodes {nodes} "
                  f"wall {dt:.3f} s ({nodes / dt / 1e9:.2f} G nodes/s)", flush=True)
            del s
