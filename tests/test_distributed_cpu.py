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


# ---- DistributedSearch (termination + hand-over) with a deterministic stand-in for the device-backed solver ----------
class _FakeProblem:
    ws, n = 2, 8


class _FakeSolver:
    """Work units are integers; processing unit u at step time spawns children 2u+1, 2u+2 while u < limit -- a binary tree
    with a known node count.  The incumbent is the largest unit seen.  Only rank 0 starts with the root: every other rank
    depends on the hand-over to get any work at all."""

    def __init__(self, rank, limit, batch):
        self.problem = _FakeProblem()
        self.open = [0] if rank == 0 else []
        self.limit, self.batch = limit, batch
        self.lb = -(1 << 62)
        self.done = 0

    def step(self):
        if not self.open:
            return 0
        take, self.open = self.open[:self.batch], self.open[self.batch:]
        for u in take:
            self.done += 1
            self.lb = max(self.lb, u)
            if u < self.limit:
                self.open += [2 * u + 1, 2 * u + 2]
        return 1

    def epoch(self, prev, max_steps=64, min_ms=2.0):
        """what ddo_solver_epoch does natively (include/ddo_hip.h): import of the reduced incumbent, ONE step (the stand-in keeps the
        epochs short so that the hand-over logic is exercised), this rank's vector for the next MAX all-reduce"""
        low = -(1 << 62)
        if prev is not None and prev[0] > low:
            self.import_lower_bound(prev[0])
        rc = self.step()
        n = len(self.open)
        top = max(self.open) if self.open else low
        return rc, [max(self.lb, low), 1 if rc == 1 else 0, 0, n, -n, top, -top if n else low]

    def flush(self):
        return 0

    def fringe_len(self):
        return len(self.open)

    def fringe_best_ub(self):
        return max(self.open) if self.open else -(1 << 62)

    def best_lower_bound(self):
        return self.lb

    def import_lower_bound(self, lb):
        self.lb = max(self.lb, lb)

    def export_subproblems(self, k):
        import numpy as np
        take, self.open = self.open[:k], self.open[k:]
        n = len(take)
        return {"states": np.array([[u, u + 1] for u in take], dtype=np.uint64).reshape(n, 2), "value": np.array(take, dtype=np.int64),
                "ub": np.array(take, dtype=np.int64), "depth": np.zeros(n, dtype=np.int64),
                "path_off": np.arange(n + 1, dtype=np.uint64), "paths": np.array([[u, 1] for u in take], dtype=np.int64).reshape(n, 2)}

    def import_subproblems(self, nodes):
        for i, u in enumerate(nodes["value"]):
            assert int(nodes["states"][i][0]) == int(u) and int(nodes["states"][i][1]) == int(u) + 1
            lo, hi = int(nodes["path_off"][i]), int(nodes["path_off"][i + 1])
            assert hi - lo == 1 and int(nodes["paths"][lo][0]) == int(u)
            self.open.append(int(u))


def _search_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddo_amd.distributed import DistributedSearch

    s = _FakeSolver(rank, limit=2000, batch=16)
    search = DistributedSearch(s, dist, "cpu", rebalance_every=2, donate_min=4, donate_max=64)
    proved, best = search.maximize()
    q.put((rank, proved, best, s.done, search.nodes_sent, search.nodes_received, len(s.open)))
    dist.destroy_process_group()


def test_distributed_search_terminates_and_hands_work_over_world3_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_search_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the tree: units 0..limit-1 have two children each -> 2 * limit + 1 units in total, largest unit 2 * (limit - 1) + 2
    assert sum(o[3] for o in out) == 2 * 2000 + 1
    assert all(o[1] is True and o[2] == 2 * 1999 + 2 for o in out)        # every rank knows the global incumbent
    assert all(o[6] == 0 for o in out)                                     # nothing left open anywhere
    assert out[1][3] > 0 and out[2][3] > 0                                 # the ranks that started empty did get work
    assert sum(o[4] for o in out) == sum(o[5] for o in out) > 0           # every node sent was received exactly once


def _skew_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddo_amd.distributed import DistributedSearch

    # every rank has work, but rank 0 holds all the promising units (large bounds) and many of them: leaves only, nothing spawns
    s = _FakeSolver(rank, limit=0, batch=4)
    s.open = list(range(100000, 100400)) if rank == 0 else list(range(10 * rank, 10 * rank + 12))
    search = DistributedSearch(s, dist, "cpu", rebalance_every=1, donate_min=4, donate_max=64, ub_gap=2)
    proved, best = search.maximize()
    q.put((rank, proved, best, s.done, search.nodes_sent, search.nodes_received, len(s.open)))
    dist.destroy_process_group()


def test_rebalancing_by_best_open_bound_world2_gloo():
    """No rank runs dry, but one of them sits on all the promising sub-problems: with ub_gap the hand-over starts although
    every rank has open nodes, point to point (send / recv), and every unit is processed exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_skew_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(o[3] for o in out) == 400 + 12
    assert all(o[1] is True and o[2] == 100399 for o in out) and all(o[6] == 0 for o in out)
    assert out[0][4] > 0 and out[1][5] == out[0][4]        # rank 0 gave, rank 1 received the same number
    assert out[1][3] > 12                                   # ... and worked on them


class _StingySolver(_FakeSolver):
    """exports at most ONE unit whatever it is asked for: with two receivers one share is empty"""

    def export_subproblems(self, k):
        return super().export_subproblems(min(k, 1))


def _empty_share_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddo_amd.distributed import DistributedSearch

    s = _StingySolver(rank, limit=0, batch=2)          # leaves only: nothing spawns
    s.open = list(range(50, 90)) if rank == 0 else []  # ranks 1 and 2 are dry receivers
    search = DistributedSearch(s, dist, "cpu", rebalance_every=1, donate_min=2, donate_max=8)
    proved, best = search.maximize()
    q.put((rank, proved, best, s.done, search.nodes_sent, search.nodes_received, len(s.open), search.handovers))
    dist.destroy_process_group()


def test_a_receiver_whose_share_is_empty_world3_gloo():
    """The donor exports fewer nodes than there are receivers: the receiver of the empty share gets a header and
    nothing else, does not count a hand-over, and the search still processes every unit exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_share_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(o[3] for o in out) == 40 and all(o[1] is True and o[2] == 89 for o in out) and all(o[6] == 0 for o in out)
    assert out[0][4] == out[1][5] + out[2][5] > 0          # sent == received, one unit per hand-over
    assert out[2][5] == 0 and out[2][7] == 0               # the second receiver's share was always empty: no hand-over counted there
    assert out[1][5] == out[1][7]                          # the first receiver: one unit per counted hand-over
