"""CPU suite, part 4: the N > 1 plumbing (ddo_amd/distributed.py) with world_size 2 on the gloo backend.
The data path itself needs no collective: ranks own disjoint fringe shards and exchange only the incumbent."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddo_amd.distributed import exchange_incumbent, open_work, reduce_stats

    # each rank improves its own incumbent at different steps; everyone must see the running maximum
    local = [[-(1 << 63), 20, 20, 23], [-(1 << 63), 18, 24, 24]][rank]
    seen = []
    lb = -(1 << 63)
    for step in range(4):
        lb = max(lb, local[step])
        lb = exchange_incumbent(dist, lb, "cpu")
        seen.append(lb)
    from ddo_amd.distributed import PipelinedIncumbent
    pipe, lb2, seen2 = PipelinedIncumbent(dist, "cpu"), -(1 << 63), []
    for step in range(4):
        lb2 = max(lb2, local[step])
        got = pipe.post(lb2)            # the global bound of the previous post
        seen2.append(got)
        if got is not None:
            lb2 = max(lb2, got)
    seen2.append(pipe.drain())
    assert pipe.drain() is None
    elapsed, sums = reduce_stats(dist, 1.0 + rank, [10 * (rank + 1), 3], "cpu")
    total_open = open_work(dist, [5, 0][rank], "cpu")
    done = open_work(dist, 0, "cpu")
    q.put((rank, seen, elapsed, sums, total_open, done, seen2))
    dist.destroy_process_group()


def test_incumbent_exchange_and_stats_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, elapsed, sums, total_open, done, seen2 in out:
        assert seen == [-(1 << 62), 20, 24, 24]
        assert seen2 == [None, -(1 << 62), 20, 24, 24]      # one step stale, same running maximum
        assert elapsed == 2.0 and sums == [30.0, 6.0]
        assert total_open == 5 and done == 0


def test_single_process_identity():
    from ddo_amd.distributed import exchange_incumbent, open_work, reduce_stats

    assert exchange_incumbent(None, 17, "cpu") == 17
    from ddo_amd.distributed import PipelinedIncumbent
    p = PipelinedIncumbent(None, "cpu")
    assert p.post(5) == 5 and p.drain() is None
    assert reduce_stats(None, 0.5, [1, 2], "cpu") == (0.5, [1.0, 2.0])
    assert open_work(None, 9, "cpu") == 9
