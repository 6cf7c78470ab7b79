#!/usr/bin/env python3
"""The CPU oracle's two DD types side by side on the MISP instances of examples/misp/tests.rs: sequential B&B, NbUnassignedWidth,
default DD (clean.rs) against the pooled one (pooled.rs: long arcs over the variables that do not impact a node).  One JSON line
per instance and DD type -> profiles/r0N/pooled_oracle.jsonl.  CPU only (the device has no pooled variant).

    python tools/pooled_vs_default.py [seconds per search = 20]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.oracle_binding import Oracle  # noqa: E402

INSTANCES = {"c-fat200-1": 12, "c-fat200-2": 24, "c-fat200-5": 58, "c-fat500-1": 14, "c-fat500-2": 26, "hamming6-2": 32, "hamming6-4": 4,
             "hamming8-2": 128, "johnson8-2-4": 4, "johnson8-4-4": 14, "MANN_a9": 16, "brock200_2": 12, "p_hat300-1": 8}
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
for name, expected in INSTANCES.items():
    inst = o.misp(os.path.join(ROOT, "data", "misp", name + ".clq"))
    for pooled in (False, True):
        t0 = time.time()
        r = inst.solve(0, 0, budget, pooled=pooled)
        print(json.dumps({"instance": name, "dd": "pooled" if pooled else "default", "expected": expected, "best_value": r["best_value"],
                          "proved": bool(r["is_exact"]), "explored": r["explored"], "nodes_expanded": r["nodes_expanded"], "arcs": r["arcs"],
                          "compiles": r["compiles"], "wall_s": round(time.time() - t0, 3), "budget_s": budget}), flush=True)
