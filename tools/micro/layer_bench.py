#!/usr/bin/env python3
"""SURVEY.md section 8 d2 micro-benchmark: decision-diagram compilation of SYNTHETIC MISP sub-problems, no branch-and-bound.

    gpurun -- python tools/micro/layer_bench.py [--quick]

Grid: n in {200, 400} x edge probability p in {0.25, 0.50} (graph G(n, p), seeds 1-3: brock400_1's complement has density
0.25, brock200_2's 0.50) x width W in {1 000, 10 000, 100 000} x batch B in {1, 16, 256} sub-problems per launch.  A
sub-problem is a random vertex subset (uniform bits of density 0.5 ANDed with a random mask, seed 1), value 0, depth 0;
each one gets a RESTRICTED and a RELAXED compile through ddo_mdd_compile_batch (the C ABI: one launch per batch).
Reported per cell: nodes expanded per second of kernel... of wall time of the batch call, the algorithmic GB/s
((S + 8) + c (S + 16) bytes per node, S = 8 ceil(n / 64)) and its fraction of the 8 TB/s HBM roofline, and which device
engine served the width (in-place layers up to W = 32 767, the layer-rebuilding engine above: its dedup table does not fit
the LDS).  One JSON line per cell; `--quick` runs one seed; W = 100 000 x B = 256 only with --huge."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddo_amd  # noqa: E402
from ddo_amd import CompilationType, SubProblem  # noqa: E402


def gnp_rows(n, p, seed):
    """complement-adjacency rows of G(n, p): bit j of row i set <=> i and j are NOT adjacent (compatible), i != j"""
    rng = np.random.RandomState(seed)
    adj = np.triu(rng.rand(n, n) < p, 1)
    adj = adj | adj.T
    ws = (n + 63) // 64
    rows = np.zeros((n, ws), dtype=np.uint64)
    for i in range(n):
        for j in np.nonzero(~adj[i])[0]:
            if j != i:
                rows[i, j // 64] |= np.uint64(1) << np.uint64(j % 64)
    return rows.reshape(-1)


def random_states(n, count, seed=1):
    rng = np.random.RandomState(seed)
    ws = (n + 63) // 64
    out = []
    for _ in range(count):
        bits = (rng.rand(n) < 0.5) & (rng.rand(n) < 0.9)
        s = np.zeros(ws, dtype=np.uint64)
        for i in np.nonzero(bits)[0]:
            s[i // 64] |= np.uint64(1) << np.uint64(i % 64)
        out.append(s)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--huge", action="store_true", help="also W = 100 000 x B = 256")
    args = ap.parse_args()
    seeds = [1] if args.quick else [1, 2, 3]
    for n in (200, 400):
        for p in (0.25, 0.50):
            for seed in seeds:
                model = ddo_amd.Misp.from_rows(n, gnp_rows(n, p, seed), np.ones(n, dtype=np.int64))
                for W in (1000, 10000, 100000):
                    for B in (1, 16, 256):
                        if W == 100000 and B == 256 and not args.huge:
                            continue          # 256 cut-sets of 100 000 nodes overflow the shared 1 GB output arena: the compiles are
                                              # repeated one by one (minutes per cell); --huge runs the cell anyway
                        states = random_states(n, B)
                        try:
                            mdds = [ddo_amd.Mdd(model, W) for _ in range(B)]
                        except ddo_amd.DdoError as e:
                            print(json.dumps({"n": n, "p": p, "seed": seed, "W": W, "B": B, "error": str(e)[:120]}), flush=True)
                            continue
                        subs = [SubProblem(state=s, value=0, path=[], depth=0) for s in states]
                        tot_nodes = tot_arcs = 0
                        t0 = time.perf_counter()
                        for ct in (CompilationType.Restricted, CompilationType.Relaxed):
                            ddo_amd.Mdd.compile_batch(mdds, [ct] * B, [W] * B, subs, [-(1 << 40)] * B)
                            for m in mdds:
                                c = m.counters()
                                tot_nodes += c["nodes_expanded"]
                                tot_arcs += c["arcs"]
                        dt = time.perf_counter() - t0
                        S = 8 * ((n + 63) // 64)
                        cmean = tot_arcs / max(1, tot_nodes)
                        bpn = (S + 8) + cmean * (S + 16)
                        gbs = tot_nodes * bpn / dt / 1e9
                        print(json.dumps({"n": n, "p": p, "seed": seed, "W": W, "B": B, "engine": "in-place" if W < 32767 else "rebuild",
                                          "nodes": tot_nodes, "wall_ms": 1e3 * dt, "nodes_per_s": tot_nodes / dt, "bytes_per_node": bpn,
                                          "GBps": gbs, "hbm_frac": gbs / 8000.0}), flush=True)
                        del mdds


if __name__ == "__main__":
    main()
