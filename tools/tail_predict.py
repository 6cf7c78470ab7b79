#!/usr/bin/env python3
"""How uneven are the sub-problems of one launch, and what could a longest-first order buy?

Root DDs of brock400_1 at W = 10 000 (restricted, then relaxed: its cut-set is the fringe), every `stride`-th cut-set node in
MaxUB order up to `count`; each is compiled alone (restricted + relaxed, the incumbent of the root) and its nodes recorded next
to what the host knows before the launch: popcount of the state, value, ub.  Then list scheduling of the sizes on `slots`
workgroups is simulated: in fringe order (what the engine does today), longest-first by each predictor, longest-first by the
true size (the bound).  One JSON line per sub-problem on stdout, the summary on stderr.

    python tools/tail_predict.py [count=1024] [stride=8] [slots=512]
"""
import heapq
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ddo_amd
from ddo_amd import SubProblem
from ddo_amd.binding import CompilationType


def makespan(sizes, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for s in sizes:
        heapq.heappush(h, heapq.heappop(h) + s)
    return max(h)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    slots = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    W = 10000
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = ddo_amd.Misp.read_instance(os.path.join(root, "data", "misp", "brock400_1.clq"))
    mdd = ddo_amd.Mdd(model, W)
    r = model.root()
    lb = -(1 << 60)
    c = mdd.compile(CompilationType.Restricted, W, r, lb)
    lb = max(lb, c.best_value)
    mdd.compile(CompilationType.Relaxed, W, r, lb)
    cut = mdd.drain_cutset()
    cut.sort(key=lambda s: -s.ub)          # MaxUB (stable: ties keep the cut-set order)
    picked = cut[::stride][:count]
    rows = []
    for i, sp in enumerate(picked):
        sub = SubProblem(state=sp.state, value=sp.value, path=[], ub=sp.ub, depth=sp.depth)
        nodes = 0
        cr = mdd.compile(CompilationType.Restricted, W, sub, lb)
        nodes += mdd.counters()["nodes_expanded"]
        nr = nodes
        if not cr.is_exact:
            mdd.compile(CompilationType.Relaxed, W, sub, lb)
            nodes += mdd.counters()["nodes_expanded"]
        pop = int(sum(bin(int(w)).count("1") for w in sp.state))
        rows.append({"i": i, "popcount": pop, "value": int(sp.value), "ub": int(sp.ub), "depth": int(sp.depth), "nodes": int(nodes), "nodes_restricted": int(nr)})
        print(json.dumps(rows[-1]))
    n = np.array([r["nodes"] for r in rows], dtype=np.float64)
    mean_load = n.sum() / slots
    res = {"count": len(rows), "slots": slots, "nodes_total": float(n.sum()), "largest": float(n.max()), "mean": float(n.mean()), "p50": float(np.median(n)),
           "p90": float(np.percentile(n, 90)), "mean_load_per_slot": mean_load}
    res["makespan_fringe_order"] = makespan(n, slots) / mean_load
    res["makespan_lpt_true"] = makespan(sorted(n, reverse=True), slots) / mean_load
    for key in ("popcount", "ub", "value"):
        f = np.array([r[key] for r in rows], dtype=np.float64)
        order = np.argsort(-f, kind="stable")
        res["makespan_lpt_" + key] = makespan(n[order], slots) / mean_load
        res["corr_" + key] = float(np.corrcoef(f, n)[0, 1])
    f = np.array([r["ub"] - r["value"] for r in rows], dtype=np.float64)
    res["makespan_lpt_ub_minus_value"] = makespan(n[np.argsort(-f, kind="stable")], slots) / mean_load
    res["corr_ub_minus_value"] = float(np.corrcoef(f, n)[0, 1])
    print(json.dumps(res), file=sys.stderr)


if __name__ == "__main__":
    main()
