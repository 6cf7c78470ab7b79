"""The reference's HOST over the device, measured: the oracle's restatement of ParallelSolver (worker threads, one mutex-shared
critical section, NoDupFringe of host-resident states and paths) with `HipMdd` as its DecisionDiagram (tests/shim/hip_mdd_shim.cpp),
on the headline instance under a time budget.  What a Rust user of hip_mdd/ would see end to end; compare bench.py's `boundary_b1`
(the same boundary without the host's fringe) and `value` (the ready-made host whose fringe stays in HBM).
    python tools/shim_bench.py [instance=brock400_1] [width=10000] [seconds=20] [threads...=64 512 2048]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.shim_binding import shim_misp_solve   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "brock400_1"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
threads = [int(x) for x in sys.argv[4:]] or [64, 512, 2048]
for t in threads:
    r = shim_misp_solve(os.path.join(ROOT, "data", "misp", name + ".clq"), width, t, timeout_s=seconds)
    print(json.dumps({"instance": name, "width": width, "threads": t, "budget_s": seconds, "wall_s": r["wall_s"], "proved": bool(r["is_exact"]),
                      "best_lb": r["best_lb"], "best_ub": r["best_ub"], "explored": r["explored"], "compiles": r["compiles"],
                      "nodes_expanded": r["nodes_expanded"], "nodes_per_s": r["nodes_expanded"] / max(r["wall_s"], 1e-9),
                      "launches": r["launches"], "decision_diagrams_per_launch": r["requests"] / max(1, r["launches"])}), flush=True)
