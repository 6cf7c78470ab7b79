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
    """`ub_gap`: a hand-over is also started when the best open bound of some rank lies at least that far below the best
    open bound of another one (0 = only when a rank runs dry): hash-sharded best-first searches each follow their LOCAL
    order, and a rank left with unpromising nodes expands sub-problems a global best-first search would have reached much
    later, or never (search overhead).  The donor's best nodes then go round."""

    def __init__(self, solver, dist, device, rebalance_every=4, donate_min=64, donate_max=32768, ub_gap=2, epoch_ms=2.0, max_steps=64):
        self.s, self.dist, self.device = solver, dist, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.rebalance_every, self.donate_min, self.donate_max, self.ub_gap = rebalance_every, donate_min, donate_max, ub_gap
        self.epoch_ms, self.max_steps = epoch_ms, max_steps   # an epoch: search steps for this long (at least one, at most max_steps)
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

    def _to_dev(self, arr):
        a = np.ascontiguousarray(arr)
        return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(self.device)

    def _send_nodes(self, nodes, idx, dst, ws):
        """point-to-point: the records `idx` of an export as a two-word header and -- unless the share is empty -- ONE packed
        int64 message (states, value, ub, depth, path offsets, paths): xGMI is point-to-point, a receiver gets its own share and
        nothing else, and a share costs two transfers instead of seven"""
        k = len(idx)
        offs = nodes["path_off"].astype(np.int64)
        lens = offs[idx + 1] - offs[idx] if k else np.zeros(0, dtype=np.int64)
        new_off = np.zeros(k + 1, dtype=np.int64)
        new_off[1:] = np.cumsum(lens)
        tot = int(new_off[-1])
        self.dist.send(self._to_dev(np.array([k, tot], dtype=np.int64)), dst=dst)
        if k == 0:
            return
        paths = np.concatenate([nodes["paths"][offs[i]:offs[i + 1]] for i in idx]).reshape(-1, 2) if tot else np.zeros((0, 2), dtype=np.int64)
        parts = [nodes["states"][idx].reshape(k, ws).view(np.int64).ravel(), nodes["value"][idx].astype(np.int64), nodes["ub"][idx].astype(np.int64),
                 nodes["depth"][idx].astype(np.int64), new_off, paths.astype(np.int64).ravel()]
        self.dist.send(self._to_dev(np.concatenate(parts)), dst=dst)

    def _recv_nodes(self, src, ws):
        hdr = torch.zeros(2, dtype=torch.int64, device=self.device)
        self.dist.recv(hdr, src=src)
        k, tot = int(hdr[0].item()), int(hdr[1].item())
        if k == 0:   # an empty share: nothing else was sent
            return {"states": np.zeros((0, ws), dtype=np.uint64), "value": np.zeros(0, dtype=np.int64), "ub": np.zeros(0, dtype=np.int64),
                    "depth": np.zeros(0, dtype=np.int64), "path_off": np.zeros(1, dtype=np.uint64), "paths": np.zeros((0, 2), dtype=np.int64)}
        t = torch.zeros(k * ws + 3 * k + (k + 1) + 2 * tot, dtype=torch.int64, device=self.device)
        self.dist.recv(t, src=src)
        a = t.cpu().numpy()
        o = 0
        out = {}
        for name, n, shape, dt in (("states", k * ws, (k, ws), np.uint64), ("value", k, (k,), np.int64), ("ub", k, (k,), np.int64),
                                   ("depth", k, (k,), np.int64), ("path_off", k + 1, (k + 1,), np.uint64), ("paths", 2 * tot, (tot, 2), np.int64)):
            piece = a[o:o + n].reshape(shape)
            out[name] = piece.view(np.uint64) if dt == np.uint64 else piece
            o += n
        return out

    def _handover(self, ws):
        """All ranks call this together.  Donor = the rank with the best open bound (most open nodes among equals); receivers =
        the ranks with no open node, and -- with ub_gap -- the ranks whose best open bound lies ub_gap below the donor's and
        that hold far fewer nodes.  Each receiver gets its interleaved share of the donor's best nodes, point to point."""
        dist = self.dist
        n_open = int(self.s.fringe_len())
        top = int(self.s.fringe_best_ub()) if n_open > 0 else I64_LOW
        mine = torch.tensor([n_open, max(top, I64_LOW)], dtype=torch.int64, device=self.device)
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine)
        counts = [int(c[0].item()) for c in allc]
        tops = [int(c[1].item()) for c in allc]
        donor = max(range(self.world), key=lambda r: (tops[r] if counts[r] >= self.donate_min else I64_LOW, counts[r], -r))
        receivers = [r for r in range(self.world) if r != donor and
                     (counts[r] == 0 or (self.ub_gap > 0 and tops[donor] - tops[r] >= self.ub_gap and 4 * counts[r] < counts[donor]))]
        if not receivers or counts[donor] < self.donate_min * (len(receivers) + 1):
            return False
        share = min(self.donate_max, counts[donor] // (len(receivers) + 1))
        if self.rank == donor:
            nodes = self.s.export_subproblems(share * len(receivers))
            k = len(nodes["value"])
            for j, r in enumerate(receivers):
                self._send_nodes(nodes, np.arange(j, k, len(receivers)), r, ws)   # interleaved: everyone gets nodes from the top of the donor's order
            self.handovers += 1 if k else 0
            self.nodes_sent += k
            return k > 0
        if self.rank in receivers:
            got = self._recv_nodes(donor, ws)
            n = len(got["value"])
            if n:
                self.s.import_subproblems(got)
                self.handovers += 1
            self.nodes_received += n
            return n > 0          # (a receiver whose share was empty has no new work: it does not step on nothing)
        return False

    def maximize(self):
        """Runs the sharded search to completion on every rank.  Returns (is_exact, global best value or None)."""
        s = self.s
        if self.dist is None:
            c = s.maximize()
            return bool(c.is_exact), c.best_value
        ws = s.problem.ws
        aborted = False
        local_work = True
        prev = None
        while True:
            # one epoch = ONE native call (ddo_solver_epoch: import of the reduced incumbent, search steps for `epoch_ms` milliseconds, this
            # rank's vector for the next reduction) and ONE asynchronous MAX all-reduce, consumed an epoch later.  Round 4 paced the loop
            # in Python, a collective per step: small instances spent their time in it (profiles/r04/dist_overhead.jsonl).
            if local_work or s.fringe_len() > 0:
                rc, vec = s.epoch(prev, self.max_steps, self.epoch_ms)
            else:
                if prev is not None and prev[0] > I64_LOW:
                    s.import_lower_bound(prev[0])
                rc, vec = 0, [max(int(s.best_lower_bound()), I64_LOW), 0, 0, 0, 0, I64_LOW, I64_LOW]
            if rc == DDO_CUTOFF:
                aborted = True
            local_work = rc == 1
            self.epochs += 1
            vec[2] = 1 if aborted else vec[2]
            prev = self._post(vec)
            if prev is None:
                continue
            lb, any_work, any_abort, max_open, neg_min_open, max_top, neg_min_top = prev   # (the incumbent is imported by the next epoch)
            if any_abort:        # a time budget ran out somewhere: everybody stops (parallel.rs:479-489 abort_search)
                aborted = True
                break
            if not any_work:
                break
            dry = -neg_min_open == 0 and max_open >= 2 * self.donate_min
            skew = self.ub_gap > 0 and neg_min_top > I64_LOW and max_top + neg_min_top >= self.ub_gap and max_open >= 4 * self.donate_min
            # (a dry rank is served at the next rebalancing epoch, a mere skew of the bounds four times less often: a hand-over
            # costs an export, a transfer and an import -- profiles/r03/dist_overhead.jsonl)
            if (dry and self.epochs % self.rebalance_every == 0) or (skew and self.epochs % (4 * self.rebalance_every) == 0):
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
