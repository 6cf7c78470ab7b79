#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X:

    MDD nodes expanded / second, MISP on DIMACS brock400_1, width 10 000   (BASELINE.json, config C4)

The timed workload is FROZEN, so the number does not depend on --steps / --warmup:

  setup   the branch-and-bound search (parallel.rs:391-437 per sub-problem: restricted DD, then -- when inexact --
          relaxed DD, both compiled on the GPU) runs its root step: the cut-set of the root's relaxed DD (<= 10 000
          nodes of ONE layer, all of them width-saturating sub-problems) is the initial fringe.  Every GPU freezes
          `--concurrent` (1024) of them together with the incumbent (ddo_solver_bench_freeze): with N ranks the root
          cut-set is sharded by state hash, and a rank keeps every (8/N)-th node of its shard in fringe order (MaxUB),
          so that for N = 1, 2, 4, 8 each GPU works on a statistically identical 1-in-8 sample of the same 8192+
          sub-problems (weak scaling: per-GPU work fixed).  The residual states already sit in the node pool (HBM).
          Later batches of a live search are NOT like these: a best-first search soon runs on small sub-problems
          (61 % of all nodes of the whole search sit in DDs that fill the width, but 97 % of the DDs do not), which
          is why the whole search is reported separately (`proof`, with its own roofline figure).
  step    one pass of the hot path over one frozen batch: batch (k mod --batches) is compiled exactly as a search step
          would (same launch, same software pipeline, same host work on the (ub, value) rows coming back), but nothing
          is folded into the fringe.  Every cycle of `--batches` steps is therefore the same work.

`value` = nodes expanded in the K timed steps / wall time of those steps (whole job, all ranks).  The kernel-only rate
(HIP events on the engine's own stream) feeds `roofline`.  After the timed steps, at N = 1 only and outside the timed
region: a bounded CPU sample (`cpu_baseline`) and a fresh whole search to the PROVED optimum
(`time_to_proved_optimum_s`, the second half of BASELINE.json's metric, with its own roofline figure).

`python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run -- every rank owns a shard of the root cut-set (hash of the state) and only the incumbent lower
bound crosses GPUs (one 8-byte MAX all-reduce over RCCL/xGMI per step).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INSTANCE = "brock400_1"
WIDTH = 10000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PREFIX_STEPS = 0               # search steps between the root step and the frozen batches (fixed: part of the workload)


def physical_cores():
    """(physical cores, logical cpus) of this host from /proc/cpuinfo."""
    cores = set()
    phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return (len(cores) or (os.cpu_count() or 1)), (os.cpu_count() or 1)


def kernel_source_fingerprint():
    """sha256 (16 hex digits) over the device / host sources the product library is built from: ties a rocprof summary to the
    build it was collected on (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ddo_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hpp", ".h", ".hip", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(key):
    """HBM bytes per expanded node of a secondary workload from the newest committed rocprofv3 --pmc passes (profiles/<round>/
    pmc_extra.json: FETCH_SIZE x calibrated factor + WRITE_SIZE over the nodes expanded under the counters -- tools/profile_round.sh,
    tools/summarize_extra.py), or (None, reason).  Counters of another build of the kernels say nothing about this run."""
    fp = kernel_source_fingerprint()
    pdir = os.path.join(ROOT, "profiles")
    for rnd in sorted((d for d in os.listdir(pdir) if d.startswith("r")), reverse=True) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, rnd, "pmc_extra.json")
        if not os.path.exists(f):
            continue
        e = json.load(open(f)).get("secondary", {}).get(key)
        if not e or "hbm_bytes_per_node" not in e:
            return None, f"null: profiles/{rnd}/pmc_extra.json holds no counter pass of {key}"
        if e.get("kernel_sources") != fp:
            return None, f"null: profiles/{rnd}/pmc_extra.json ({key}) was collected on kernel sources {e.get('kernel_sources')}, this build is {fp}"
        return e["hbm_bytes_per_node"], (f"profiles/{rnd}/pmc_extra.json: {key} (FETCH_SIZE x {e.get('fetch_factor')} [calibrated] + WRITE_SIZE over the "
                                         f"{e.get('nodes'):.0f} nodes expanded under the counters, kernel sources {fp}) x nodes per launch")
    return None, "null: no profiles/<round>/pmc_extra.json"


def cpu_baseline(instance, width, total_seconds, threads_arg):
    """The CPU oracle (C++ restatement of ddo, kind = "port") on the same instance / width, built -O3 -march=native for
    the host this runs on, swept over thread counts inside a bounded time budget; the best configuration is reported."""
    from tests.oracle_binding import Oracle

    lib = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    build = "-O3 -march=x86-64-v2 (prebuilt)"
    try:   # the prebuilt library is portable x86-64-v2: rebuild the same source for this host, outside the tree
        out = os.path.join(tempfile.gettempdir(), "ddo_oracle_native_%d" % os.getuid(), "liboracle_native.so")
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "NATIVE_OUT=" + out], check=True, timeout=300,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib, build = out, "-O3 -march=native (built on this host)"
    except Exception:
        pass
    o = Oracle(lib)
    inst = o.misp(os.path.join(ROOT, "data", "misp", instance + ".clq"))
    phys, logical = physical_cores()
    # the root sub-problem (two 400-layer DDs of width 10 000) is compiled by ONE thread before the others find work: samples
    # shorter than about 10 s mostly measure that ramp-up, so two thread counts share the budget (round 1: 32 was the best of
    # 1/32/128/256, 64 the runner-up)
    sweep = [threads_arg] if threads_arg > 0 else sorted({t for t in (32, 64) if t <= logical}) or [logical]
    per = max(2.0, total_seconds / len(sweep))
    runs = []
    for t in sweep:
        r = inst.solve(width, t, per)
        runs.append({"threads": t, "nodes_per_s": r["nodes_expanded"] / max(r["wall_s"], 1e-9), "subproblems": r["explored"],
                     "nodes": r["nodes_expanded"], "wall_s": r["wall_s"]})
    best = max(runs, key=lambda x: x["nodes_per_s"])
    return {
        "value": best["nodes_per_s"], "unit": "nodes/s", "cores": best["threads"], "kind": "port",
        "sample": f"oracle ParallelSolver (C++ restatement of ddo, {build}), same instance/width, TimeBudget {per:.0f} s per thread "
                  f"count: best = {best['threads']} threads, {best['subproblems']} sub-problems, {best['nodes']} nodes in {best['wall_s']:.1f} s",
        "physical_cores": phys, "logical_cpus": logical, "thread_sweep": runs,
    }


def boundary_b1(model, args, device, threads, rounds):
    """Throughput THROUGH THE DROP-IN BOUNDARY as the reference drives it: `threads` host threads, one
    ddo_mdd each, every thread looping the reference's process_one_node (parallel.rs:391-437: restricted compile, then -- the
    restricted DD being inexact -- relaxed compile + drain_cutset) through plain `ddo_mdd_compile` (tools/b1_driver.cpp).  The
    sub-problems are the same kind of 1-in-8 sample of the root cut-set the headline freezes (exported to the HOST here: through
    this boundary states, cut-sets and paths cross PCIe per compile, which `value` never pays -- its fringe stays in HBM).
    Concurrent compiles share device launches (Engine::compile_combined)."""
    import numpy as np

    import ddo_amd
    from ddo_amd import FixedWidth, ParallelSolver
    from ddo_amd.boundary import run_b1

    tmp = ParallelSolver(model, FixedWidth(args.width), nb_threads=args.concurrent, device=device, fringe="lazy")
    tmp.step()
    tmp.flush()
    lb = tmp.best_lower_bound()
    ex = tmp.export_subproblems(8 * args.concurrent)
    del tmp
    k = len(ex["value"])
    pick = list(range(0, k, 8))[:args.concurrent]
    states = np.ascontiguousarray(ex["states"][pick])
    values = [int(ex["value"][i]) for i in pick]
    depths = [int(ex["depth"][i]) for i in pick]
    keep = ddo_amd.Mdd(model, args.width, device=device)   # keeps the engine of this (model, device, width) alive between the two passes
    run_b1(model, states, values, depths, args.width, threads, len(pick), lb, device=device)             # untimed: buffers pinned, arenas sized
    tot, r0, r1 = run_b1(model, states, values, depths, args.width, threads, rounds * len(pick), lb, device=device)
    del keep
    return {
        "value": tot["nodes_expanded"] / max(tot["seconds"], 1e-9), "unit": "nodes/s", "threads": threads,
        "what": "T host threads, one ddo_mdd each, looping process_one_node (parallel.rs:391-437) through plain ddo_mdd_compile / "
                "ddo_mdd_drain_cutset (tools/b1_driver.cpp); concurrent compiles share device launches",
        "sample": f"{len(pick)} sub-problems of the root cut-set (every 8th in fringe order, states on the host), {rounds} passes: "
                  f"{tot['compiles']} compiles, {tot['nodes_expanded']} nodes in {tot['seconds']:.2f} s",
        "launches": tot["launches"], "compiles": tot["compiles"], "decision_diagrams_per_launch": tot["requests"] / max(1, tot["launches"]),
        "kernel_s": tot["kernel_ms"] / 1e3, "kernel_nodes_per_s": tot["nodes_expanded"] / max(tot["kernel_ms"] / 1e3, 1e-9),
        "cutset_nodes_drained": tot["cutset_nodes"], "path_decisions_drained": tot["path_decisions"], "errors": tot["errors"],
        "results_differing_between_passes": tot["mismatches"],
    }


def bench_vector(args):
    """Secondary workloads on ONE GPU, the whole branch-and-bound to the proved optimum: `max2sat` = BASELINE.json config C3,
    weighted MAX2SAT frb10-6-1 (n = 60, 667 clauses) at FixedWidth(5000); `mcp` = the ten maximum-cut instances the reference's
    tests solve (examples/mcp/tests.rs:64-103, n = 30) at FixedWidth(100), one search after the other.  The signed-vector models
    run on the layer-rebuilding engine (misp_dd_core.hpp) with the host NoDupFringe: `value` = nodes expanded / wall time of
    maximize(); the kernel-only rate (HIP events on the engine's stream: the DELTA over this search -- solvers of one model share
    their engine) feeds `roofline` with SURVEY.md section 8 d3's figure for these models, S = 4 n + 4 bytes and c = 2 children:
    bytes_per_node = (S + 8) + 2 (S + 16)."""
    import ddo_amd
    from ddo_amd import FixedWidth, ParallelSolver
    from tests.oracle_binding import Oracle

    budget = None
    if args.workload == "max2sat":
        width = 5000
        name = args.instance if args.instance != INSTANCE else "frb10-6-1"     # e.g. --instance frb15-9-1 (n = 135: not proved within minutes,
        cases = [(os.path.join(ROOT, "data", "max2sat", name + ".wcnf"), None)]  # the search then runs under --prove SECONDS and reports its gap)
        load, label = ddo_amd.Max2Sat.read_instance, f"MAX2SAT {name}.wcnf"
        if name != "frb10-6-1":
            budget = ddo_amd.TimeBudget(min(args.prove, 60.0) if args.prove > 0 else 30.0)
    else:
        width = 100
        optima = [13, 14, 13, 12, 13, 15, 15, 15, 13, 13]   # filled from the oracle below when it disagrees (it never has)
        cases = [(os.path.join(ROOT, "data", "mcp", f"mcp_n30_p0.1_{i:03d}.mcp"), None) for i in range(10)]
        load, label = ddo_amd.Mcp.read_instance, "MCP mcp_n30_p0.1_000..009.mcp"
    models = [load(p) for p, _ in cases]
    tot = None
    nodes_all = 0            # nodes expanded by BOTH passes: what a profiler wrapped around this process has counted
    for rep in range(2):     # first pass = warm-up (engine creation, first launches)
        tot = {"dt": 0.0, "nodes": 0, "kms": 0.0, "launches": 0, "explored": 0, "compiles": 0, "values": [], "proved": True}
        for model in models:
            s = ParallelSolver(model, FixedWidth(width), budget if rep == 1 else (ddo_amd.TimeBudget(2.0) if budget else None),
                               nb_threads=args.concurrent, fringe="nodup")
            k0, l0 = s.device_time()
            t0 = time.perf_counter()
            c = s.maximize()
            dt = time.perf_counter() - t0
            k1, l1 = s.device_time()
            cnt = s.counters()
            tot["dt"] += dt
            tot["nodes"] += cnt["nodes_expanded"]
            nodes_all += cnt["nodes_expanded"]
            tot["kms"] += k1 - k0
            tot["launches"] += l1 - l0
            tot["explored"] += s.explored()
            tot["compiles"] += cnt["compiles"]
            tot["values"].append(c.best_value)
            tot["proved"] = tot["proved"] and bool(c.is_exact)
            tot.setdefault("bounds", []).append((s.best_lower_bound(), s.best_upper_bound(), s.gap()))
            del s
    assert tot["kms"] / 1e3 <= tot["dt"] * 1.001, "kernel time must fit inside the wall time of the searches it belongs to"
    n = models[0].n
    S = 4 * n + 4
    bpn = (S + 8) + 2 * (S + 16)
    ach = tot["nodes"] * bpn / max(tot["kms"] / 1e3, 1e-12) / 1e9
    tkey = ("max2sat_" + name.replace("-", "_")) if args.workload == "max2sat" else "mcp_n30"
    tpn, tsrc = committed_traffic(tkey)
    out = {
        "metric": f"MDD nodes expanded/sec, {label} w={width} (whole search to the proved optimum)",
        "value": tot["nodes"] / tot["dt"], "unit": "nodes/s", "n_gpus": 1, "steps": int(tot["launches"]), "warmup": 1,
        "ms_per_step": 1e3 * tot["dt"] / max(1, tot["launches"]), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32",
        "data": "real instances shipped with the reference (resources/max2sat, resources/mcp)",
        "config": {"workload": f"{label} FixedWidth({width}) LEL cut-set, EmptyCache, host NoDupFringe(MaxUB), "
                               f"{args.concurrent} sub-problems per launch", "parallelism": "1 GPU"},
        "proved": tot["proved"], "best_values": tot["values"], "time_to_proved_optimum_s": tot["dt"] if tot["proved"] else None,
        "bounds_lb_ub_gap": tot.get("bounds"), "time_budget_s": budget.seconds if budget else None,
        "subproblems": tot["explored"], "compiles": tot["compiles"],
        "profile_key": tkey, "kernel_sources": kernel_source_fingerprint(), "nodes_all_passes": nodes_all,
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": None if tpn is None else tpn * tot["nodes"] / max(1, tot["launches"]), "traffic_unit": "bytes per launch",
                     "traffic_source": tsrc,
                     "kernel": "ddo_hip::misp_compile_kernel<WS, table in LDS> (layer-rebuilding engine, signed-vector states: "
                               f"{(n + 1) // 2 + 1} words)",
                     "kernel_ms_avg": tot["kms"] / max(1, tot["launches"]), "kernel_s": tot["kms"] / 1e3, "wall_s": tot["dt"],
                     "launches": int(tot["launches"]), "bytes_per_node": bpn,
                     "kernel_nodes_per_s": tot["nodes"] / max(tot["kms"] / 1e3, 1e-12)},
    }
    if not args.no_cpu:
        o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
        phys, logical = physical_cores()
        runs = []
        for t in (1, 16, 32):
            nodes, wall, ok = 0, 0.0, True
            for (path, opt), got in zip(cases, tot["values"]):
                if args.workload == "max2sat":
                    v, r = o.max2sat_file(path, width, t, args.cpu_seconds / 3)
                else:
                    v, r = o.mcp_file(path, width, t)
                nodes += r["nodes_expanded"]
                wall += r["wall_s"]
                ok = ok and (not r["is_exact"] or v == got)    # same proved optimum as the device search
            runs.append({"threads": t, "nodes_per_s": nodes / max(wall, 1e-9), "wall_s": wall, "same_optimum": ok})
        best = max(runs, key=lambda x: x["nodes_per_s"])
        out["cpu_baseline"] = {"value": best["nodes_per_s"], "unit": "nodes/s", "cores": best["threads"], "kind": "port",
                               "sample": "oracle ParallelSolver (C++ restatement of ddo), same instances/width" +
                                         (f", TimeBudget {args.cpu_seconds / 3:.0f} s per thread count" if args.workload == "max2sat" else ", whole searches"),
                               "physical_cores": phys, "logical_cpus": logical, "thread_sweep": runs}
        out["speedup_vs_cpu"] = out["value"] / max(best["nodes_per_s"], 1e-9)
    print(json.dumps(out), flush=True)


def bench_tsptw(args):
    """Secondary workload on ONE GPU: BASELINE.json config C5 -- TSPTW, resources/tsptw n ~ 40, width 20000 -- as whole searches
    to the proved optimum in the reference's example configuration (examples/tsptw/main.rs:70-128: frontier cut-set, SimpleCache,
    SimpleDominanceChecker(TsptwDominance)) with FixedWidth(20000) in place of TsptwWidth; `--instance FAMILY/FILE` runs one
    instance of data/tsptw under TsptwWidth(nb_vars, 1) instead (e.g. AFG/rbg125a.tw: 126 nodes, 2-word node sets).
    D-ary layer-rebuilding engine (misp_dd_core.hpp + dd_tsptw.hpp), host NoDupFringe.  `roofline` uses SURVEY.md section 8 d3's
    formula with the TSPTW state (S = 8 * state words) and the MEASURED mean fan-out c = arcs / nodes expanded:
    bytes_per_node = (S + 8) + c (S + 16)."""
    import ddo_amd
    from ddo_amd import FRONTIER, FixedWidth, ParallelSolver, TsptwWidth
    from tests.oracle_binding import Oracle

    if args.instance != INSTANCE:
        cases = [os.path.join(ROOT, "data", "tsptw", args.instance)]
        width, wlabel, label = TsptwWidth(1), "TsptwWidth(nb_vars, 1)", f"TSPTW {args.instance}"
    else:
        cases = [os.path.join(ROOT, "data", "tsptw", "Langevin", f + ".dat") for f in ("N40ft201", "N40ft207", "N40ft403", "N40ft410")]
        width, wlabel, label = FixedWidth(20000), "FixedWidth(20000)", "TSPTW Langevin N40ft201/207/403/410 (config C5)"
    models = [ddo_amd.Tsptw.read_instance(p) for p in cases]
    tot = None
    nodes_all = 0            # nodes expanded by BOTH passes: what a profiler wrapped around this process has counted
    for rep in range(2):     # first pass = warm-up (engine creation, first launches)
        tot = {"dt": 0.0, "nodes": 0, "arcs": 0, "kms": 0.0, "launches": 0, "explored": 0, "compiles": 0, "values": [], "proved": True}
        for model in models:
            s = ParallelSolver(model, width, ddo_amd.TimeBudget(120.0), nb_threads=args.concurrent, fringe="nodup", cutset_type=FRONTIER,
                               cache_entries=1 << 22, dominance_entries=1 << 22)
            k0, l0 = s.device_time()
            t0 = time.perf_counter()
            c = s.maximize()
            dt = time.perf_counter() - t0
            k1, l1 = s.device_time()
            cnt = s.counters()
            tot["dt"] += dt
            tot["nodes"] += cnt["nodes_expanded"]
            nodes_all += cnt["nodes_expanded"]
            tot["arcs"] += cnt["arcs"]
            tot["kms"] += k1 - k0
            tot["launches"] += l1 - l0
            tot["explored"] += s.explored()
            tot["compiles"] += cnt["compiles"]
            tot["values"].append(c.best_value)
            tot["proved"] = tot["proved"] and bool(c.is_exact)
            del s
    assert tot["kms"] / 1e3 <= tot["dt"] * 1.001, "kernel time must fit inside the wall time of the searches it belongs to"
    S = 8 * models[0].ws
    tkey = "tsptw_c5" if args.instance == INSTANCE else "tsptw_" + os.path.basename(args.instance).split(".")[0]
    tpn, tsrc = committed_traffic(tkey)
    fan = tot["arcs"] / max(1, tot["nodes"])
    bpn = (S + 8) + fan * (S + 16)
    ach = tot["nodes"] * bpn / max(tot["kms"] / 1e3, 1e-12) / 1e9
    out = {
        "metric": f"MDD nodes expanded/sec, {label} {wlabel} (whole searches to the proved optimum)",
        "value": tot["nodes"] / tot["dt"], "unit": "nodes/s", "n_gpus": 1, "steps": int(tot["launches"]), "warmup": 1,
        "ms_per_step": 1e3 * tot["dt"] / max(1, tot["launches"]), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 sets / i32 times",
        "data": "real instances shipped with the reference (resources/tsptw)",
        "config": {"workload": f"{label} {wlabel} frontier cut-set, SimpleCache, TsptwDominance, host NoDupFringe(MaxUB), "
                               f"{args.concurrent} sub-problems in flight", "parallelism": "1 GPU"},
        "proved": tot["proved"], "best_values": tot["values"], "time_to_proved_optimum_s": tot["dt"] if tot["proved"] else None,
        "subproblems": tot["explored"], "compiles": tot["compiles"], "arcs_per_node": fan,
        "profile_key": tkey, "kernel_sources": kernel_source_fingerprint(), "nodes_all_passes": nodes_all,
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": None if tpn is None else tpn * tot["nodes"] / max(1, tot["launches"]), "traffic_unit": "bytes per launch",
                     "traffic_source": tsrc,
                     "kernel": f"ddo_hip::misp_compile_kernel<WS, table in LDS / HBM> (layer-rebuilding engine, {models[0].ws}-word TSPTW states, "
                               f"{models[0].n} children per node)",
                     "kernel_ms_avg": tot["kms"] / max(1, tot["launches"]), "kernel_s": tot["kms"] / 1e3, "wall_s": tot["dt"],
                     "launches": int(tot["launches"]), "bytes_per_node": bpn,
                     "kernel_nodes_per_s": tot["nodes"] / max(tot["kms"] / 1e3, 1e-12)},
    }
    if not args.no_cpu:
        o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
        phys, logical = physical_cores()
        runs = []
        if args.instance != INSTANCE:      # the oracle's ParallelSolver in the same configuration, swept over thread counts
            for t in (1, 16, 32):
                v, r = o.tsptw_file(cases[0], 1, t)
                runs.append({"threads": t, "nodes_per_s": r["nodes_expanded"] / max(r["wall_s"], 1e-9), "wall_s": r["wall_s"],
                             "same_optimum": (not r["is_exact"]) or v == tot["values"][0]})
        else:                               # fixed width: the oracle's traced sequential search (one thread)
            nodes, wall, ok = 0, 0.0, True
            for path, got in zip(cases, tot["values"]):
                r, _ = o.trace_ex("tsptw+dominance", path, 20000, 0, True, True)
                nodes += r["nodes_expanded"]
                wall += r["wall_s"]
                ok = ok and r["best_value"] == got
            runs.append({"threads": 1, "nodes_per_s": nodes / max(wall, 1e-9), "wall_s": wall, "same_optimum": ok})
        best = max(runs, key=lambda x: x["nodes_per_s"])
        out["cpu_baseline"] = {"value": best["nodes_per_s"], "unit": "nodes/s", "cores": best["threads"], "kind": "port",
                               "sample": "oracle (C++ restatement of ddo), same instances / width / cut-set / cache / dominance, whole searches",
                               "physical_cores": phys, "logical_cpus": logical, "thread_sweep": runs}
        out["speedup_vs_cpu"] = out["value"] / max(best["nodes_per_s"], 1e-9)
    print(json.dumps(out), flush=True)


def bench_pooled(args):
    """Secondary line on ONE GPU: the reference's second decision-diagram type, `Pooled` (mdd/pooled.rs), beside the
    default one on the same instances -- whole searches under the reference's own MISP configuration (NbUnassignedWidth, NoDupFringe, MaxUB,
    examples/misp/main.rs) and a time budget: brock200_4 and the headline instance brock400_1; `--instance NAME` runs one other DIMACS graph
    of data/misp.  Per (instance, DD type): nodes expanded per second, sub-problems explored, compiles, the bounds reached.  `value` = the
    pooled search's rate on the first instance; the pooled kernel is the in-place engine with POOLED = 1, one 512-thread workgroup per CU."""
    import ddo_amd
    from ddo_amd import NbUnassignedWidth, ParallelSolver

    names = [args.instance] if args.instance != INSTANCE else ["brock200_4", "brock400_1"]
    budget_s = min(args.prove, 30.0) if args.prove > 0 else 20.0
    rows = []
    for name in names:
        model = ddo_amd.Misp.read_instance(os.path.join(ROOT, "data", "misp", name + ".clq"))
        for dd in ("default", "pooled", "pooled+cache"):
            for rep in range(2):   # first pass = warm-up (engine creation), 2 s
                s = ParallelSolver(model, NbUnassignedWidth(model.n), ddo_amd.TimeBudget(budget_s if rep else 2.0), nb_threads=args.concurrent, fringe="nodup",
                                   pooled=dd != "default", cache_entries=(1 << 22) if dd == "pooled+cache" else 0)
                k0, l0 = s.device_time()
                t0 = time.perf_counter()
                c = s.maximize()
                dt = time.perf_counter() - t0
                k1, l1 = s.device_time()
                cnt = s.counters()
                row = {"instance": name, "dd": dd, "proved": bool(c.is_exact), "best_value": c.best_value, "best_lb": s.best_lower_bound(),
                       "best_ub": s.best_upper_bound(), "wall_s": dt, "subproblems": s.explored(), "compiles": cnt["compiles"],
                       "nodes_expanded": cnt["nodes_expanded"], "arcs": cnt["arcs"], "nodes_per_s": cnt["nodes_expanded"] / max(dt, 1e-9),
                       "kernel_s": (k1 - k0) / 1e3, "launches": int(l1 - l0),
                       "kernel_nodes_per_s": cnt["nodes_expanded"] / max((k1 - k0) / 1e3, 1e-9)}
                del s
            rows.append(row)
    first = next(r for r in rows if r["dd"] == "pooled")
    base = next(r for r in rows if r["dd"] == "default" and r["instance"] == first["instance"])
    n = ddo_amd.Misp.read_instance(os.path.join(ROOT, "data", "misp", first["instance"] + ".clq")).n
    S = 8 * ((n + 63) // 64)
    cmean = first["arcs"] / max(1, first["nodes_expanded"])
    bpn = (S + 8) + cmean * (S + 16)
    ach = first["nodes_expanded"] * bpn / max(first["kernel_s"], 1e-12) / 1e9
    print(json.dumps({
        "metric": f"MDD nodes expanded/sec, MISP {first['instance']} Pooled decision diagrams, NbUnassignedWidth (search under a time budget)",
        "value": first["nodes_per_s"], "unit": "nodes/s", "n_gpus": 1, "steps": first["launches"], "warmup": 1,
        "ms_per_step": 1e3 * first["wall_s"] / max(1, first["launches"]), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "real instances (DIMACS graphs shipped with the reference)",
        "config": {"workload": f"MISP {', '.join(names)}: ParallelSolver over Pooled / Pooled + SimpleCache / default decision diagrams, NbUnassignedWidth, "
                               f"NoDupFringe(MaxUB), {args.concurrent} sub-problems per launch, TimeBudget {budget_s:.0f} s", "parallelism": "1 GPU"},
        "time_budget_s": budget_s, "searches": rows, "pooled_vs_default_nodes_per_s": first["nodes_per_s"] / max(base["nodes_per_s"], 1e-9),
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                     "traffic_source": "null: no rocprofv3 --pmc pass of this workload is committed",
                     "kernel": "ddo_hip::misp_compile_kernel2_pooled<WS> (in-place engine, POOLED = 1, 512 threads, one decision diagram per CU)",
                     "kernel_s": first["kernel_s"], "launches": first["launches"], "bytes_per_node": bpn, "kernel_nodes_per_s": first["kernel_nodes_per_s"]},
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="misp", choices=["misp", "max2sat", "mcp", "tsptw", "misp-pooled"],
                    help="misp: the headline metric (default); max2sat: BASELINE config C3 on one GPU (secondary line); tsptw: config C5 on one GPU")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--concurrent", type=int, default=1024, help="sub-problems compiled per step and GPU (== the reference's nb_threads)")
    ap.add_argument("--batches", type=int, default=1, help="frozen batches the timed steps cycle through")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="total time budget of the CPU baseline sample (split over the thread sweep)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = 32 and 64, report the best)")
    ap.add_argument("--no-cpu", action="store_true", help="timed steps only: neither the CPU baseline nor the proof search")
    ap.add_argument("--prove", type=float, default=400.0, metavar="SECONDS",
                    help="after the timed steps (N = 1 only), run the whole search to the PROVED optimum under this time budget and report "
                         "time_to_proved_optimum_s, the second half of BASELINE.json's metric (0 = skip)")
    ap.add_argument("--prove-concurrent", type=int, default=32768,
                    help="sub-problems in flight during the proof search (8192: 93 s, 16384: 83 s, 32768: 78 s, 65536+: 79 s on one box, round 3)")
    ap.add_argument("--b1-threads", type=int, default=2048, help="host threads of the boundary_b1 block (N = 1 only; 0 = skip): worker threads looping plain ddo_mdd_compile")
    ap.add_argument("--b1-rounds", type=int, default=16, help="passes of the boundary_b1 block over its sub-problems")
    ap.add_argument("--freeze-stride", type=int, default=0, help="experiments only: take every k-th root cut-set node (default 8 // world)")
    ap.add_argument("--instance", default=INSTANCE)
    ap.add_argument("--width", type=int, default=WIDTH)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if args.workload in ("max2sat", "mcp"):
        if args.concurrent == 1024:
            args.concurrent = 256
        return bench_vector(args)
    if args.workload == "tsptw":
        if args.concurrent == 1024:
            args.concurrent = 64
        return bench_tsptw(args)
    if args.workload == "misp-pooled":
        if args.concurrent == 1024:
            args.concurrent = 256
        return bench_pooled(args)
    # test hook (single-GPU boxes): DDO_BENCH_ONE_GPU=1 puts every rank on cuda:0 and rendezvous over gloo, so the
    # multi-process path (sharded root cut-set, incumbent all-reduce, max-over-ranks timing) can be exercised there
    one_gpu = os.environ.get("DDO_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm_device = "cpu" if one_gpu else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist

        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import ddo_amd
    from ddo_amd import FixedWidth, ParallelSolver

    model = ddo_amd.Misp.read_instance(os.path.join(ROOT, "data", "misp", args.instance + ".clq"))
    conc = args.concurrent
    solver = ParallelSolver(model, FixedWidth(args.width), nb_threads=conc, device=local_rank, rank=rank, world_size=world,
                            fringe="lazy")

    from ddo_amd.distributed import PipelinedIncumbent, reduce_stats

    incumbent = PipelinedIncumbent(dist, comm_device)

    def barrier():
        solver.flush()   # the engine runs on its own HIP stream: wait for the launch in flight and absorb it
        if dist is not None:
            lb = incumbent.drain()   # the all-reduce in flight belongs to the steps before the barrier
            if lb is not None:
                solver.import_lower_bound(lb)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def exchange():
        if dist is not None:   # parallel.rs:439-453: the incumbent is the only datum shared between workers
            lb = incumbent.post(solver.best_lower_bound())   # one step stale: ranks do not run in lock-step
            if lb is not None:
                solver.import_lower_bound(lb)

    # ---- setup (untimed): root sub-problem, fixed prefix of the search, then freeze the workload
    solver.step()
    for _ in range(PREFIX_STEPS):
        solver.step()
        exchange()
    barrier()
    stride = args.freeze_stride or max(1, 8 // world)
    nfrozen = solver.bench_freeze(args.batches, stride)
    if nfrozen < 1:
        raise SystemExit("bench.py: the fringe ran dry before the workload could be frozen")

    def one_step():
        solver.bench_step()
        exchange()

    for _ in range(args.warmup):
        one_step()
    barrier()
    c0 = solver.counters()
    k0, l0 = solver.device_time()
    ts0 = solver.tier_stats()
    e0 = solver.explored()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    t1 = time.perf_counter()
    c1 = solver.counters()
    k1, l1 = solver.device_time()
    ts1 = solver.tier_stats()
    e1 = solver.explored()
    done_steps = args.steps

    elapsed, (nodes, arcs, subs, compiles) = reduce_stats(
        dist, t1 - t0, [c1["nodes_expanded"] - c0["nodes_expanded"], c1["arcs"] - c0["arcs"], e1 - e0,
                        c1["compiles"] - c0["compiles"]], comm_device)

    if rank == 0:
        ws_bytes = 8 * ((model.n + 63) // 64)                    # S: state bytes
        my_nodes = c1["nodes_expanded"] - c0["nodes_expanded"]
        my_arcs = c1["arcs"] - c0["arcs"]
        c_mean = my_arcs / max(1, my_nodes)                       # mean children per expanded node
        bytes_per_node = (ws_bytes + 8) + c_mean * (ws_bytes + 16)
        ws_t = next(w for w in (1, 2, 4, 7, 8, 16) if w >= (model.n + 63) // 64)
        # the sub-problems of a step run on the engine tier that fits them (host_solver.cpp: dispatch): the DOMINANT kernel is
        # the tier with the most kernel time over the timed region; its roofline uses the nodes IT expanded and ITS launches
        tiers = []
        for a, b in zip(ts0, ts1):
            name = (f"ddo_hip::misp_compile_kernel2_dense<{ws_t}> (512 threads, two decision diagrams per CU)" if b["dense"]
                    else f"ddo_hip::misp_compile_kernel2<{ws_t}, {b['threads']}>" if b is ts1[-1]
                    else f"ddo_hip::misp_compile_kernel2_tier<{ws_t}> ({b['threads']} threads, layer capacity {b['layer_capacity']})")
            tiers.append({"kernel": name, "kernel_ms": b["kernel_ms"] - a["kernel_ms"], "launches": b["launches"] - a["launches"],
                          "subproblems": b["subproblems"] - a["subproblems"], "handed_up": b["retried"] - a["retried"],
                          "nodes_expanded": b["nodes_expanded"] - a["nodes_expanded"], "slots": b["slots"], "lds_bytes": b["lds_bytes"]})
        dom = max(tiers, key=lambda t: t["kernel_ms"]) if tiers else None
        if dom and dom["kernel_ms"] > 0 and dom["launches"] > 0:
            launches, kernel_s, dom_nodes, dom_name = dom["launches"], dom["kernel_ms"] / 1e3, dom["nodes_expanded"], dom["kernel"]
        else:
            launches, kernel_s, dom_nodes, dom_name = max(1, l1 - l0), (k1 - k0) / 1e3, my_nodes, f"ddo_hip::misp_compile_kernel2<{ws_t}, 1024>"
        achieved = dom_nodes * bytes_per_node / max(kernel_s, 1e-12) / 1e9   # GB/s over the kernel's own time
        # HBM traffic: PMC counters cannot be read from inside this process; the committed rocprofv3 passes of this very
        # command and workload (tools/profile_round.sh -> profiles/<round>/pmc_summary.json; the workload is frozen, so
        # the profiled launches ARE these launches) give bytes per expanded node, scaled by the nodes of one launch.
        # The summary is stamped with a fingerprint of the kernel sources and the dominant kernel's name: counters of another
        # build, or of another kernel, say nothing about this run -- traffic is then null, with the reason.
        traffic, traffic_src = None, "no profiles/<round>/pmc_summary.json"
        pdir = os.path.join(ROOT, "profiles")
        for rnd in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
            pj = os.path.join(pdir, rnd, "pmc_summary.json")
            if os.path.exists(pj):
                try:
                    sj = json.load(open(pj))
                    tj = sj.get("traffic")
                    if tj and tj.get("hbm_bytes_per_node"):
                        stamp = sj.get("stamp") or {}
                        if stamp.get("kernel_sources") != kernel_source_fingerprint():
                            traffic_src = (f"null: profiles/{rnd}/pmc_summary.json was collected on kernel sources {stamp.get('kernel_sources')}, "
                                           f"this build is {kernel_source_fingerprint()} (re-run tools/profile_round.sh)")
                        elif stamp.get("kernel") != dom_name:
                            traffic_src = f"null: profiles/{rnd}/pmc_summary.json describes {stamp.get('kernel')}, the dominant kernel here is {dom_name}"
                        else:
                            traffic = tj["hbm_bytes_per_node"] * dom_nodes / launches
                            traffic_src = (f"profiles/{rnd}/pmc_summary.json (FETCH_SIZE x calibrated factor + WRITE_SIZE per node -- counter_calibration.json --, same frozen workload, kernel sources "
                                           f"{stamp.get('kernel_sources')}, git {stamp.get('git')}) x nodes per launch")
                        break
                except (ValueError, OSError):
                    pass
        out = {
            "metric": "MDD nodes expanded/sec, MISP brock400_1 w=10k (restricted+relaxed DD compilation inside B&B)",
            "value": nodes / elapsed,
            "unit": "nodes/s",
            "n_gpus": world,
            "steps": done_steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, done_steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "dtype_note": "states are u64 bit-set words; values, bounds and ranking keys are 32-bit integers on the device "
                          "(key32 = (value - vbase) << 11 | popcount), bit-exact against the i64 oracle for sum|w| < 2^20",
            "data": "real instance (DIMACS brock400_1 complement graph shipped with the reference); frozen batches of the live search",
            "config": {"workload": f"MISP {args.instance}.clq FixedWidth({args.width}) LEL cut-set, EmptyCache, SimpleFringe(MaxUB) kept in "
                                   f"the device node pool; frozen workload per GPU: {solver.bench_frozen()} sub-problems of the root cut-set "
                                   f"(every {stride}-th node of the rank's shard in fringe order), {nfrozen} batch(es), cycled",
                       "dense_tier": ("on" if os.environ.get("DDO_HIP_DENSE", "1") != "0" else "off") +
                                     " (default on: two 512-thread decision diagrams per CU; round 3, one box: on 1.703e10 nodes/s / 30.6 % of the"
                                     " roofline, off -- DDO_HIP_DENSE=0, the 1024-thread kernel, one per CU -- 1.552e10 / 27.9 %)",
                       "launch_order": ("input order (DDO_HIP_LPT=0)" if os.environ.get("DDO_HIP_LPT", "1") == "0" else
                                        "the decision diagrams of a launch are drawn longest first: sorted on the device by the vertices left in "
                                        "their residual states (round 4, one box, alternating: 1.80-1.84e10 nodes/s against 1.64-1.68e10 in the "
                                        "host's order)"),
                       "subproblems_per_step": conc, "frozen_batches": nfrozen, "prefix_steps": PREFIX_STEPS,
                       "parallelism": f"fringe-shard x{world}"},
            "subproblems_per_s": subs / elapsed,
            "compiles": compiles,
            "best_lb": solver.best_lower_bound(),
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "kernel": dom_name, "kernel_sources": kernel_source_fingerprint(), "kernel_ms_avg": 1e3 * kernel_s / launches, "launches": launches,
                "bytes_per_node": bytes_per_node, "children_per_node": c_mean, "nodes_per_launch": dom_nodes / launches,
                "kernel_nodes_per_s": dom_nodes / max(kernel_s, 1e-12),
                "all_kernels": {"kernel_ms": k1 - k0, "nodes_expanded": my_nodes, "GBps": my_nodes * bytes_per_node / max((k1 - k0) / 1e3, 1e-12) / 1e9},
                "tiers": tiers,
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.instance, args.width, args.cpu_seconds, args.cpu_threads)
            out["speedup_vs_cpu"] = out["value"] / max(out["cpu_baseline"]["value"], 1e-9)
    else:
        out = None
    # ---- secondary metric of BASELINE.json: wall time of maximize() to the PROVED optimum, on
    # all N GPUs: every rank searches its shard of the root cut-set; incumbent, termination test and work hand-over go
    # through ddo_amd.distributed.DistributedSearch (one 40-byte MAX all-reduce per step)
    proof = None
    if world == 1 and not args.no_cpu and args.b1_threads > 0:
        del solver
        solver = None
        out["boundary_b1"] = boundary_b1(model, args, local_rank, args.b1_threads, args.b1_rounds)
        out["boundary_b1"]["fraction_of_value"] = out["boundary_b1"]["value"] / max(out["value"], 1e-9)
    if args.prove > 0 and not args.no_cpu:   # --no-cpu = the timed steps only (profiling, A/B tools)
        from ddo_amd import TimeBudget
        from ddo_amd.distributed import DistributedSearch
        del solver
        prover = ParallelSolver(model, FixedWidth(args.width), TimeBudget(args.prove), nb_threads=args.prove_concurrent, device=local_rank,
                                rank=rank, world_size=world, fringe="lazy")
        search = DistributedSearch(prover, dist, comm_device)
        if dist is not None:
            dist.barrier()
        tp0 = time.perf_counter()
        proved, best = search.maximize()
        tp = time.perf_counter() - tp0
        pc = prover.counters()
        pk_ms, pl = prover.device_time()
        tp, (p_nodes, p_arcs, p_subs, p_kms, p_sent) = reduce_stats(
            dist, tp, [pc["nodes_expanded"], pc["arcs"], prover.explored(), pk_ms, search.nodes_sent], comm_device)
        proof = (proved, best, tp, p_nodes, p_arcs, p_subs, p_kms, p_sent, pl, prover.best_upper_bound())
        proof_tiers = [{"layer_capacity": t["layer_capacity"], "threads": t["threads"], "dense": bool(t["dense"]), "slots": t["slots"],
                        "kernel_s": t["kernel_ms"] / 1e3, "launches": t["launches"], "subproblems": t["subproblems"], "handed_up": t["retried"],
                        "nodes_expanded": t["nodes_expanded"]} for t in prover.tier_stats()]
    if rank == 0:
        if proof is not None:
            proved, best, tp, p_nodes, p_arcs, p_subs, p_kms, p_sent, pl, p_ub = proof
            ws_bytes = 8 * ((model.n + 63) // 64)
            p_c = p_arcs / max(1.0, p_nodes)
            p_bpn = (ws_bytes + 8) + p_c * (ws_bytes + 16)
            p_ach = p_nodes * p_bpn / max(p_kms / 1e3 / world, 1e-12) / 1e9 / world    # per GPU: kernel seconds are summed over ranks
            out["time_to_proved_optimum_s"] = tp if proved else None
            out["proof"] = {"proved": bool(proved), "best_value": best, "wall_s": tp, "subproblems": int(p_subs), "nodes_expanded": int(p_nodes),
                            "budget_s": args.prove, "subproblems_in_flight_per_gpu": args.prove_concurrent, "n_gpus": world,
                            "nodes_per_s": p_nodes / max(tp, 1e-9), "subproblems_handed_over": int(p_sent),
                            "roofline": {"bound": "hbm", "achieved": p_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU", "frac": p_ach / HBM_PEAK_GBS,
                                         "kernel_s_sum_over_gpus": p_kms / 1e3, "launches_rank0": pl, "bytes_per_node": p_bpn,
                                         "note": "every compile launch of the whole search, all capacity tiers (their kernel time, HIP events)"},
                            "tiers_rank0": proof_tiers}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
