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
