#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X:

    MDD nodes expanded / second, MISP on DIMACS brock400_1, width 10 000   (BASELINE.json, config C4)

One *step* = one round of the branch-and-bound host over the device engine (the root sub-problem is expanded during
setup: it is a batch of one): up to `--concurrent`
sub-problems are popped from the fringe and each gets its restricted and (when inexact) relaxed
decision diagram compiled on the GPU in a single launch (parallel.rs:391-437), then their cut-sets
are enqueued.  Inputs are resident in HBM when the timed region of the kernel starts (the
sub-problem states are 56-byte records uploaded before the launch; the PCIe-inclusive wall rate is
what `value` reports, the kernel-only rate is in `roofline`).

`python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run -- every rank owns a shard of the root cut-set and only the incumbent lower
bound crosses GPUs (one 8-byte MAX all-reduce over RCCL/xGMI per step).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INSTANCE = "brock400_1"
WIDTH = 10000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(seconds, threads):
    """The CPU oracle (C++ restatement of ddo, kind = "port") on the same instance / width for a bounded
    time budget; nodes expanded per second on `threads` host threads."""
    from tests.oracle_binding import Oracle

    o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    inst = o.misp(os.path.join(ROOT, "data", "misp", INSTANCE + ".clq"))
    r = inst.solve(WIDTH, threads, seconds)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--concurrent", type=int, default=2048, help="sub-problems compiled per step (== the reference's nb_threads)")
    ap.add_argument("--fringe", default="lazy", choices=["lazy", "nodup"],
                    help="lazy: cut-sets stay in the device node pool, SimpleFringe/MaxUB order; nodup: host NoDupFringe")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=32, help="threads of the CPU baseline (32 = best of the 1/32/128/256 sweep on the GPU box; 0 = all)")
    ap.add_argument("--no-cpu", action="store_true", help="timed steps only: neither the CPU baseline nor the proof search")
    ap.add_argument("--prove", type=float, default=400.0, metavar="SECONDS",
                    help="after the timed steps (N = 1 only), run the whole search to the PROVED optimum under this time budget and report "
                         "time_to_proved_optimum_s, the second half of BASELINE.json's metric (about 140 s for the default workload; 0 = skip)")
    ap.add_argument("--prove-concurrent", type=int, default=8192, help="sub-problems in flight during the proof search")
    ap.add_argument("--instance", default=INSTANCE)
    ap.add_argument("--width", type=int, default=WIDTH)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
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
                            fringe=args.fringe)

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

    from ddo_amd.distributed import PipelinedIncumbent, reduce_stats

    incumbent = PipelinedIncumbent(dist, comm_device)

    def one_step():
        rc = solver.step()
        if dist is not None:   # parallel.rs:439-453: the incumbent is the only datum shared between workers
            lb = incumbent.post(solver.best_lower_bound())   # one step stale: ranks do not run in lock-step
            if lb is not None:
                solver.import_lower_bound(lb)
        return rc

    # Setup, outside warm-up and timing: the first step of a search compiles the root sub-problem alone (one
    # workgroup); its cut-set is the initial fringe every later batch is popped from.
    solver.step()
    for _ in range(args.warmup):
        one_step()
    barrier()
    c0 = solver.counters()
    k0, l0 = solver.device_time()
    e0 = solver.explored()
    t0 = time.perf_counter()
    done_steps = 0
    for _ in range(args.steps):
        one_step()
        done_steps += 1
    barrier()
    t1 = time.perf_counter()
    c1 = solver.counters()
    k1, l1 = solver.device_time()
    e1 = solver.explored()

    elapsed, (nodes, arcs, subs, compiles) = reduce_stats(
        dist, t1 - t0, [c1["nodes_expanded"] - c0["nodes_expanded"], c1["arcs"] - c0["arcs"], e1 - e0,
                        c1["compiles"] - c0["compiles"]], comm_device)

    if rank == 0:
        ws_bytes = 8 * ((model.n + 63) // 64)                    # S: state bytes
        my_nodes = c1["nodes_expanded"] - c0["nodes_expanded"]
        my_arcs = c1["arcs"] - c0["arcs"]
        c_mean = my_arcs / max(1, my_nodes)                       # mean children per expanded node
        bytes_per_node = (ws_bytes + 8) + c_mean * (ws_bytes + 16)
        launches = max(1, l1 - l0)
        kernel_s = (k1 - k0) / 1e3
        ws_t = next(w for w in (1, 2, 4, 7, 8, 16) if w >= (model.n + 63) // 64)
        achieved = my_nodes * bytes_per_node / max(kernel_s, 1e-12) / 1e9   # GB/s over the kernel's own time
        # HBM traffic: PMC counters cannot be read from inside this process; the committed rocprofv3 passes of this
        # very command (tools/profile_round.sh -> profiles/<round>/pmc_summary.json) give bytes per expanded node,
        # scaled here by the nodes one launch of THIS run processed.
        traffic, traffic_src = None, None
        for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
            pj = os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")
            if os.path.exists(pj):
                try:
                    tj = json.load(open(pj)).get("traffic")
                    if tj and tj.get("hbm_bytes_per_node"):
                        traffic = tj["hbm_bytes_per_node"] * my_nodes / launches
                        traffic_src = f"profiles/{rnd}/pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE per node) x nodes of this run's launches"
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
            "data": "real instance (DIMACS brock400_1 complement graph shipped with the reference); search state synthetic-free",
            "config": {"workload": f"MISP {args.instance}.clq FixedWidth({args.width}) LEL cut-set, EmptyCache, "
                                   + ("SimpleFringe(MaxUB) kept in the device node pool" if args.fringe == "lazy" else "NoDupFringe(MaxUB) on the host"),
                       "subproblems_per_step": conc, "parallelism": f"fringe-shard x{world}"},
            "subproblems_per_s": subs / elapsed,
            "compiles": compiles,
            "best_lb": solver.best_lower_bound(),
            "fringe_len_rank0": solver.fringe_len(),
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "kernel": f"ddo_hip::misp_compile_kernel2<{ws_t}, 1024>", "kernel_ms_avg": 1e3 * kernel_s / launches, "launches": launches,
                "bytes_per_node": bytes_per_node, "children_per_node": c_mean, "nodes_per_launch": my_nodes / launches,
                "kernel_nodes_per_s": my_nodes / max(kernel_s, 1e-12),
            },
        }
        if world == 1 and not args.no_cpu:
            threads = min(args.cpu_threads, os.cpu_count() or 1) if args.cpu_threads > 0 else (os.cpu_count() or 1)
            r = cpu_baseline(args.cpu_seconds, threads)
            out["cpu_baseline"] = {
                "value": r["nodes_expanded"] / max(r["wall_s"], 1e-9), "unit": "nodes/s", "cores": threads, "kind": "port",
                "sample": f"oracle ParallelSolver (C++ restatement of ddo), same instance/width, TimeBudget {args.cpu_seconds:.0f} s: "
                          f"{r['explored']} sub-problems, {r['nodes_expanded']} nodes in {r['wall_s']:.1f} s",
            }
            out["speedup_vs_cpu"] = out["value"] / max(out["cpu_baseline"]["value"], 1e-9)
        if args.prove > 0 and world == 1 and not args.no_cpu:   # --no-cpu = the timed steps only (profiling, A/B tools)
            # secondary metric of BASELINE.json: wall time of maximize() to the proved optimum
            from ddo_amd import TimeBudget
            del solver
            prover = ParallelSolver(model, FixedWidth(args.width), TimeBudget(args.prove), nb_threads=args.prove_concurrent, device=local_rank,
                                    fringe=args.fringe)
            tp0 = time.perf_counter()
            comp = prover.maximize()
            tp = time.perf_counter() - tp0
            pc = prover.counters()
            out["time_to_proved_optimum_s"] = tp if comp.is_exact else None
            out["proof"] = {"proved": bool(comp.is_exact), "best_value": comp.best_value, "lower_bound": prover.best_lower_bound(),
                            "upper_bound": prover.best_upper_bound(), "wall_s": tp, "subproblems": prover.explored(),
                            "nodes_expanded": pc["nodes_expanded"], "budget_s": args.prove, "subproblems_in_flight": args.prove_concurrent}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
