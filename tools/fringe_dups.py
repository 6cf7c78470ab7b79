"""How many sub-problems of the throughput path are duplicates?  (missing #4.)  The reference's misp binary uses
NoDupFringe::new(MaxUB) (examples/misp/main.rs:351); bench.py's search uses the device-resident LazyFringe (SimpleFringe
semantics: a state reached through two relaxed DDs is explored twice).  Whole searches, both fringes, same width and the same
number of sub-problems in flight: explored sub-problems, nodes expanded, wall time; a search that does not finish inside its
budget reports where it stood (open nodes, bounds).
    gpurun -- python tools/fringe_dups.py > gpurun_out/fringe_dups.jsonl"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver

CASES = (("brock200_1", 10000, 4096, 120), ("brock200_1", 10000, 32768, 120), ("brock400_1", 10000, 32768, 100),
         ("brock200_2", 1000, 256, 60), ("brock200_4", 1000, 256, 60), ("brock200_1", 2000, 1024, 90), ("keller4", 100, 256, 60), ("p_hat300-1", 100, 256, 60))
if "--small" in sys.argv:
    CASES = CASES[3:]
for name, w, conc, budget in CASES:
    model = ddo_amd.Misp.read_instance(f"data/misp/{name}.clq")
    for fr in ("lazy", "nodup"):
        s = ParallelSolver(model, FixedWidth(w), ddo_amd.TimeBudget(budget), nb_threads=conc, fringe=fr)
        t0 = time.perf_counter()
        c = s.maximize()
        dt = time.perf_counter() - t0
        k = s.counters()
        print(json.dumps({"instance": name, "width": w, "in_flight": conc, "fringe": fr, "budget_s": budget, "proved": bool(c.is_exact), "best_value": c.best_value,
                          "best_upper_bound": s.best_upper_bound(), "explored": s.explored(), "open_at_end": s.fringe_len(),
                          "nodes_expanded": k["nodes_expanded"], "compiles": k.get("compiles"), "wall_s": round(dt, 3)}), flush=True)
        del s
