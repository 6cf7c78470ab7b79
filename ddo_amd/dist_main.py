"""One rank of a multi-GPU search to the proved optimum (any of the five model families), launched one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           -m ddo_amd.dist_main data/misp/brock400_1.clq -w 10000 [-t 8192] [-d SECONDS] [--fringe lazy|nodup]
    ... -m ddo_amd.dist_main data/tsptw/<instance> -w 1 --frontier --cache 4194304 --dominance 16384      (BASELINE config C5)

Every rank compiles the root, keeps its share of the root cut-set (hash of the state), and runs
ddo_amd.distributed.DistributedSearch.maximize(): incumbent exchange, termination test and work hand-over over
torch.distributed (RCCL over xGMI; gloo with DDO_BENCH_ONE_GPU=1, which also puts every rank on cuda:0 -- the
single-GPU test hook).  Rank 0 prints one JSON line."""
import argparse
import json
import os
import time


KINDS = {"misp": "Misp", "knapsack": "Knapsack", "max2sat": "Max2Sat", "mcp": "Mcp", "tsptw": "Tsptw"}


def guess_kind(path):
    low = path.lower()
    for ext, kind in ((".clq", "misp"), (".wcnf", "max2sat"), (".mcp", "mcp")):
        if low.endswith(ext):
            return kind
    return "tsptw" if "tsptw" in low else "knapsack"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("instance")
    ap.add_argument("--kind", default=None, choices=sorted(KINDS), help="model family (default: from the file name)")
    ap.add_argument("-w", "--width", type=int, default=None, help="FixedWidth (default NbUnassignedWidth; TSPTW: the factor of TsptwWidth)")
    ap.add_argument("--width-policy", default=None, choices=["fixed", "nb-unassigned", "tsptw"],
                    help="default: tsptw (TsptwWidth with factor -w) for TSPTW, fixed when -w is given, else nb-unassigned")
    ap.add_argument("-t", "--threads", type=int, default=8192, help="sub-problems in flight per GPU")
    ap.add_argument("-d", "--duration", type=float, default=0.0, help="time budget in seconds (0 = none)")
    ap.add_argument("--fringe", default=None, choices=["lazy", "nodup"], help="default: lazy (device node pool) for MISP, nodup otherwise")
    ap.add_argument("--frontier", action="store_true", help="frontier cut-set (DefaultMDDFC); needs --fringe nodup")
    ap.add_argument("--cache", type=int, default=0, help="entries of a SimpleCache per GPU (0 = EmptyCache); every rank owns its table")
    ap.add_argument("--dominance", type=int, default=0, help="entries per depth of a SimpleDominanceChecker per GPU (knapsack, TSPTW)")
    ap.add_argument("--no-handover", action="store_true", help="never move open sub-problems between ranks")
    ap.add_argument("--ub-gap", type=int, default=2, help="rebalance when the best open bounds of two ranks differ by this much (0 = only when a rank runs dry)")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"], help="auto: RCCL (nccl), gloo with DDO_BENCH_ONE_GPU=1")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (exercises the collective code path)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import ddo_amd
    from ddo_amd.distributed import DistributedSearch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    one_gpu = os.environ.get("DDO_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = args.backend if args.backend != "auto" else ("gloo" if one_gpu else "nccl")
    d = None
    if world > 1 or args.force_dist:
        if backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        d = dist
    comm_device = "cpu" if backend == "gloo" else "cuda"
    kind = args.kind or guess_kind(args.instance)
    model = getattr(ddo_amd, KINDS[kind]).read_instance(args.instance)
    policy = args.width_policy or ("tsptw" if kind == "tsptw" else ("fixed" if args.width else "nb-unassigned"))
    width = {"tsptw": lambda: ddo_amd.TsptwWidth(args.width or 1), "fixed": lambda: ddo_amd.FixedWidth(args.width),
             "nb-unassigned": lambda: ddo_amd.NbUnassignedWidth(model.n)}[policy]()
    fringe = args.fringe or ("lazy" if kind == "misp" and not (args.frontier or args.cache or args.dominance) else "nodup")
    cutoff = ddo_amd.TimeBudget(args.duration) if args.duration > 0 else None
    # (cache and dominance tables are per GPU: a threshold found on one rank prunes only there -- sound, less pruning than the
    # reference's single shared table; SURVEY.md section 8 e1 defers their replication)
    solver = ddo_amd.ParallelSolver(model, width, cutoff, nb_threads=args.threads, device=local_rank, rank=rank, world_size=world,
                                    fringe=fringe, cutset_type=ddo_amd.FRONTIER if args.frontier else ddo_amd.LAST_EXACT_LAYER,
                                    cache_entries=args.cache, dominance_entries=args.dominance)
    search = DistributedSearch(solver, d, comm_device, donate_min=(1 << 62) if args.no_handover else 64, ub_gap=args.ub_gap)
    t0 = time.perf_counter()
    proved, best = search.maximize()
    dt = time.perf_counter() - t0
    cnt = solver.counters()
    stats = [float(solver.explored()), float(cnt["nodes_expanded"]), float(search.nodes_sent), float(search.nodes_received)]
    if d is not None:
        t = torch.tensor(stats, dtype=torch.float64, device=comm_device)
        d.all_reduce(t, op=d.ReduceOp.SUM)
        stats = t.tolist()
        mine = torch.tensor([float(solver.explored())], dtype=torch.float64, device=comm_device)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        d.all_gather(per_rank, mine)
        per_rank = [int(x.item()) for x in per_rank]
    else:
        per_rank = [int(stats[0])]
    if rank == 0:
        print(json.dumps({"instance": os.path.basename(args.instance), "width": args.width, "n_gpus": world, "proved": bool(proved),
                          "best_value": best, "wall_s": dt, "subproblems": int(stats[0]), "nodes_expanded": int(stats[1]),
                          "subproblems_per_rank": per_rank, "handed_over": int(stats[2]), "received": int(stats[3]),
                          "epochs": search.epochs}), flush=True)
    if d is not None:
        d.barrier()
        d.destroy_process_group()


if __name__ == "__main__":
    main()
