"""torch.distributed plumbing for the multi-GPU search (one process per GPU, RCCL on ROCm, gloo on CPU).

The fringe shards are independent; the only data crossing ranks per step are
  * the incumbent lower bound  -> one int64 MAX all-reduce (parallel.rs:439-453 `best_lb`), and
  * open-work / counters        -> int64 / float64 SUM all-reduces (termination test of parallel.rs:512
                                   and whole-job throughput of bench.py).
8-byte messages: latency-bound on xGMI, so exactly one collective per step, never per layer."""
import torch

I64_LOW = -(1 << 62)


def exchange_incumbent(dist, lb, device):
    """MAX all-reduce of the best lower bound; returns the global value (dist None => identity)."""
    if dist is None:
        return lb
    buf = torch.tensor([max(int(lb), I64_LOW)], dtype=torch.int64, device=device)
    dist.all_reduce(buf, op=dist.ReduceOp.MAX)
    return int(buf.item())


class PipelinedIncumbent:
    """The same MAX all-reduce, one step stale: `post(lb)` returns the global bound of the PREVIOUS post (None the first
    time) and starts the all-reduce of `lb` without waiting for it.  A rank therefore waits for its peers' step k only
    when it finishes its own step k + 1: ranks whose batches take different times do not run in lock-step, and the
    staleness is the one the reference's racing threads have (parallel.rs:439-453 reads `best_lb` under a mutex while
    other threads are mid-compile).  Every rank must post the same number of times; `drain()` returns the last result."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.buf, self.work = None, None

    def _finish(self):
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        return int(self.buf.item())

    def post(self, lb):
        if self.dist is None:
            return lb
        prev = self._finish()
        self.buf = torch.tensor([max(int(lb), I64_LOW)], dtype=torch.int64, device=self.device)
        self.work = self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX, async_op=True)
        return prev

    def drain(self):
        return self._finish()


def reduce_stats(dist, elapsed, sums, device):
    """(max over ranks of elapsed, element-wise sum over ranks of `sums`)."""
    if dist is None:
        return float(elapsed), [float(x) for x in sums]
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in s.tolist()]


def open_work(dist, fringe_len, device):
    """Total number of open sub-problems over all ranks (0 <=> the search is complete everywhere)."""
    if dist is None:
        return int(fringe_len)
    buf = torch.tensor([int(fringe_len)], dtype=torch.int64, device=device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return int(buf.item())


# ------------------------------------------------------------------------------------------------------------------
# A distributed maximize(): the reference's workers (parallel.rs:573-602) are threads that share one fringe under a
# mutex and stop when the fringe is empty AND nobody is mid-compile (parallel.rs:500-559).  Here every worker is a
# process with its own GPU and its own shard of the fringe; per epoch ONE small MAX all-reduce carries
#     [incumbent, "I still have work", "I was cut off", open nodes, -open nodes]
# (one step stale, like PipelinedIncumbent), which gives every rank the global incumbent, the termination test
# (nobody has work) and the imbalance (max and min of the open counts).  When a rank sits idle while another holds a
# large fringe, all ranks enter a synchronous hand-over: the richest rank exports a batch of its best open
# sub-problems as self-contained records (ddo_solver_export_subproblems), broadcasts them, and the idle ranks import
# their share -- sub-problems are position independent, so no other state moves.
# ------------------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402

DDO_CUTOFF = 2


class DistributedSearch:
    def __init__(self, solver, dist, device, rebalance_every=4, donate_min=64, donate_max=32768):
        self.s, self.dist, self.device = solver, dist, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.rebalance_every, self.donate_min, self.donate_max = rebalance_every, donate_min, donate_max
        self.buf = self.work = None
        self.epochs = self.handovers = self.nodes_sent = self.nodes_received = 0

    # -- the per-epoch collective (asynchronous, consumed one epoch later)
    def _post(self, vec):
        prev = None
        if self.work is not None:
            self.work.wait()
            prev = [int(x) for x in self.buf.tolist()]
        self.buf = torch.tensor(vec, dtype=torch.int64, device=self.device)
        self.work = self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX, async_op=True)
        return prev

    def _drain(self):
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        return [int(x) for x in self.buf.tolist()]

    def _bcast(self, arr, src, dtype):
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64) if arr.dtype == np.uint64 else np.ascontiguousarray(arr)).to(self.device)
        self.dist.broadcast(t, src=src)
        out = t.cpu().numpy()
        return out.view(dtype) if dtype == np.uint64 else out

    def _handover(self, ws):
        """All ranks call this together.  Donor = the rank with the most open nodes; receivers = the ranks with none."""
        dist = self.dist
        mine = torch.tensor([int(self.s.fringe_len())], dtype=torch.int64, device=self.device)
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine)
        counts = [int(c.item()) for c in allc]
        donor = max(range(self.world), key=lambda r: (counts[r], -r))
        receivers = [r for r in range(self.world) if counts[r] == 0 and r != donor]
        if not receivers or counts[donor] < self.donate_min * (len(receivers) + 1):
            return False
        share = min(self.donate_max, counts[donor] // (len(receivers) + 1))
        want = share * len(receivers)
        if self.rank == donor:
            nodes = self.s.export_subproblems(want)
            hdr = np.array([len(nodes["value"]), len(nodes["paths"])], dtype=np.int64)
        else:
            nodes, hdr = None, np.zeros(2, dtype=np.int64)
        hdr = self._bcast(hdr, donor, np.int64)
        k, tot = int(hdr[0]), int(hdr[1])
        if k == 0:
            return False
        shapes = {"states": ((k, ws), np.uint64), "value": ((k,), np.int64), "ub": ((k,), np.int64), "depth": ((k,), np.int64),
                  "path_off": ((k + 1,), np.uint64), "paths": ((max(tot, 1), 2), np.int64)}
        got = {}
        for key, (shape, dt) in shapes.items():
            if self.rank == donor:
                a = nodes[key] if key != "paths" or tot else np.zeros((1, 2), dtype=np.int64)
            else:
                a = np.zeros(shape, dtype=dt)
            got[key] = self._bcast(a.reshape(shape), donor, dt).reshape(shape)
        if self.rank == donor:
            self.handovers += 1
            self.nodes_sent += k
        elif self.rank in receivers:
            j = receivers.index(self.rank)
            idx = np.arange(j, k, len(receivers))          # interleaved: every receiver gets nodes from the top of the donor's order
            offs = got["path_off"].astype(np.int64)
            lens = offs[idx + 1] - offs[idx]
            new_off = np.zeros(len(idx) + 1, dtype=np.uint64)
            new_off[1:] = np.cumsum(lens)
            paths = np.concatenate([got["paths"][offs[i]:offs[i + 1]] for i in idx]) if len(idx) and lens.sum() else np.zeros((0, 2), dtype=np.int64)
            self.s.import_subproblems({"states": got["states"][idx], "value": got["value"][idx], "ub": got["ub"][idx],
                                       "depth": got["depth"][idx], "path_off": new_off, "paths": paths})
            self.handovers += 1
            self.nodes_received += len(idx)
        return True

    def maximize(self):
        """Runs the sharded search to completion on every rank.  Returns (is_exact, global best value or None)."""
        s = self.s
        if self.dist is None:
            c = s.maximize()
            return bool(c.is_exact), c.best_value
        ws = s.problem.ws
        aborted = False
        local_work = True
        while True:
            rc = s.step() if local_work or s.fringe_len() > 0 else 0
            if rc == DDO_CUTOFF:
                aborted = True
            local_work = rc == 1
            self.epochs += 1
            open_n = int(s.fringe_len())
            prev = self._post([max(int(s.best_lower_bound()), I64_LOW), 1 if local_work else 0, 1 if aborted else 0, open_n, -open_n])
            if prev is None:
                continue
            lb, any_work, any_abort, max_open, neg_min_open = prev
            if lb > I64_LOW:
                s.import_lower_bound(lb)
            if any_abort:        # a time budget ran out somewhere: everybody stops (parallel.rs:479-489 abort_search)
                aborted = True
                break
            if not any_work:
                break
            if self.epochs % self.rebalance_every == 0 and -neg_min_open == 0 and max_open >= 2 * self.donate_min:
                # every rank saw the same reduced values: all of them take this branch together
                last = self._drain()
                if last is not None and last[0] > I64_LOW:
                    s.import_lower_bound(last[0])
                if self._handover(ws):
                    local_work = True
        last = self._drain()
        s.flush()
        lb = max(int(s.best_lower_bound()), last[0] if last else I64_LOW)
        buf = torch.tensor([lb, 1 if aborted else 0], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.MAX)
        best, ab = int(buf[0].item()), int(buf[1].item())
        return (not ab), (best if best > I64_LOW else None)
