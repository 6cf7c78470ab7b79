"""TSPTW instances beyond 64 nodes (node sets of 2 / 4 words, dd_tsptw.hpp) in the reference's example configuration
(examples/tsptw/main.rs:70-128: frontier cut-set + SimpleCache + TsptwDominance, TsptwWidth(nb_vars, 1)): the device search next
to the oracle's ParallelSolver on the host cores.  One JSON line per instance (profiles/r0N/tsptw_beyond_64.jsonl).

    python tools/tsptw_big.py [threads-on-the-device] [oracle-threads]
"""
import json
import os
import sys
import time

sys.path.insert(0, ".")
import ddo_amd
from ddo_amd import FRONTIER, ParallelSolver, TimeBudget, TsptwWidth
from tests.oracle_binding import Oracle

INSTANCES = ["AFG/rbg067a.tw", "Dumas/n80w20.001.txt", "AFG/rbg125a.tw", "AFG/rbg132.tw", "Dumas/n200w20.001.txt", "AFG/rbg233.tw"]   # the last two: only with the layer pools of round 4
if "--small" in sys.argv:
    INSTANCES = INSTANCES[:4]
    sys.argv.remove("--small")
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
othreads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
oracle = Oracle(os.path.join("oracle", "_build", "liboracle.so"))
for name in INSTANCES:
    path = os.path.join("data", "tsptw", name)
    model = ddo_amd.Tsptw.read_instance(path)
    try:
        s = ParallelSolver(model, TsptwWidth(1), TimeBudget(60.0), nb_threads=conc, fringe="nodup", cutset_type=FRONTIER,
                           cache_entries=1 << 22, dominance_entries=1 << 22)
    except ddo_amd.DdoError as err:
        print(json.dumps({"instance": name, "nb_nodes": model.n, "error": str(err)}), flush=True)
        continue
    t0 = time.perf_counter()
    c = s.maximize()
    dt = time.perf_counter() - t0
    k, launches = s.device_time()
    cnt = s.counters()
    t1 = time.perf_counter()
    ref = oracle.tsptw_file(path, 1, othreads)
    dt_ref = time.perf_counter() - t1
    rec = {"instance": name, "nb_nodes": model.n, "state_words": model.ws, "proved": bool(c.is_exact), "best_value": c.best_value,
           "wall_s": round(dt, 3), "kernel_s": round(k / 1e3, 3), "launches": launches, "explored": s.explored(),
           "nodes_expanded": cnt["nodes_expanded"], "compiles": cnt["compiles"], "nodes_per_s": round(cnt["nodes_expanded"] / max(dt, 1e-9)),
           "device_concurrency": conc}
    if ref is not None:
        rec["oracle"] = {"best_value": ref[0], "wall_s": round(dt_ref, 3), "threads": othreads, "explored": ref[1]["explored"],
                         "nodes_expanded": ref[1]["nodes_expanded"], "same_optimum": ref[0] == c.best_value}
    print(json.dumps(rec), flush=True)
