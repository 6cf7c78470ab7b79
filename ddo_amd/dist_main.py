"""One rank of a multi-GPU search to the proved optimum (MISP), launched one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           -m ddo_amd.dist_main data/misp/brock400_1.clq -w 10000 [-t 8192] [-d SECONDS] [--fringe lazy|nodup]

Every rank compiles the root, keeps its share of the root cut-set (hash of the state), and runs
ddo_amd.distributed.DistributedSearch.maximize(): incumbent exchange, termination test and work hand-over over
torch.distributed (RCCL over xGMI; gloo with DDO_BENCH_ONE_GPU=1, which also puts every rank on cuda:0 -- the
single-GPU test hook).  Rank 0 prints one JSON line."""
import argparse
import json
import os
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("instance")
    ap.add_argument("-w", "--width", type=int, default=None)
    ap.add_argument("-t", "--threads", type=int, default=8192, help="sub-problems in flight per GPU")
    ap.add_argument("-d", "--duration", type=float, default=0.0, help="time budget in seconds (0 = none)")
    ap.add_argument("--fringe", default="lazy", choices=["lazy", "nodup"])
    ap.add_argument("--no-handover", action="store_true", help="never move open sub-problems between ranks")
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
    d = None
    if world > 1:
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        d = dist
    model = ddo_amd.Misp.read_instance(args.instance)
    width = ddo_amd.FixedWidth(args.width) if args.width else ddo_amd.NbUnassignedWidth(model.n)
    cutoff = ddo_amd.TimeBudget(args.duration) if args.duration > 0 else None
    solver = ddo_amd.ParallelSolver(model, width, cutoff, nb_threads=args.threads, device=local_rank, rank=rank, world_size=world,
                                    fringe=args.fringe)
    search = DistributedSearch(solver, d, "cpu" if one_gpu else "cuda", donate_min=(1 << 62) if args.no_handover else 64)
    t0 = time.perf_counter()
    proved, best = search.maximize()
    dt = time.perf_counter() - t0
    cnt = solver.counters()
    stats = [float(solver.explored()), float(cnt["nodes_expanded"]), float(search.nodes_sent), float(search.nodes_received)]
    if d is not None:
        t = torch.tensor(stats, dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        d.all_reduce(t, op=d.ReduceOp.SUM)
        stats = t.tolist()
        mine = torch.tensor([float(solver.explored())], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
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
