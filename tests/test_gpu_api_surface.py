"""Entry points of include/ddo_hip.h that the parity suites do not reach: the per-compile cutoff flag, the exact
solution of a DD, set_primal, the open-bound / device-time queries."""
import ctypes as C

import numpy as np
import pytest

import ddo_amd
from ddo_amd import CompilationType, FixedWidth, ParallelSolver
from tests.conftest import data_path
from tests.parity_util import is_independent_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def brock():
    return ddo_amd.Misp.read_instance(data_path("misp", "brock200_2.clq"))


def test_compile_honours_the_cutoff_flag(brock):
    """Cutoff::must_stop polled by compile (clean.rs:352-354) -> Err(Reason::CutoffOccurred)"""
    mdd = ddo_amd.Mdd(brock, 100)
    raised = C.c_int(1)
    assert mdd.compile(CompilationType.Relaxed, 100, brock.root(), -(1 << 62), cutoff=raised) is None
    lowered = C.c_int(0)
    c = mdd.compile(CompilationType.Relaxed, 100, brock.root(), -(1 << 62), cutoff=lowered)
    assert c is not None and c.best_value >= 12


def test_a_cutoff_raised_mid_compile_interrupts_it():
    """A TimeBudget (cutoff.rs:302-323) that expires WHILE a compile runs: the reference polls Cutoff::must_stop at every
    layer (clean.rs:352), here the host watches the caller's flag during the launch and raises the device-visible one, which
    the layer loop polls.  A wide compile of brock400_1 takes tens of milliseconds; the flag goes up a fraction into it."""
    import threading
    import time
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock400_1.clq"))
    W = 30000
    mdd = ddo_amd.Mdd(model, W)
    flag = C.c_int(0)
    t0 = time.perf_counter()
    full = mdd.compile(CompilationType.Relaxed, W, model.root(), -(1 << 62), cutoff=flag)
    whole = time.perf_counter() - t0
    assert full is not None and whole > 0.02
    t = threading.Timer(whole / 10, lambda: setattr(flag, "value", 1))
    t0 = time.perf_counter()
    t.start()
    cut = mdd.compile(CompilationType.Relaxed, W, model.root(), -(1 << 62), cutoff=flag)
    dt = time.perf_counter() - t0
    t.join()
    assert cut is None, "the compile ran to the end although its cutoff flag went up a tenth into it"
    assert dt < 0.8 * whole
    flag.value = 0   # the device flag is lowered again: the next compile runs to the end
    again = mdd.compile(CompilationType.Relaxed, W, model.root(), -(1 << 62), cutoff=flag)
    assert again is not None and again.best_value == full.best_value


def test_a_raised_flag_stops_only_its_own_compile_of_a_batch():
    """Cutoff::must_stop is per compile (clean.rs:352).  Two wide compiles share one launch; the flag of the first goes up
    a fraction into it.  The device flag is per launch, so both DDs are cut on the device -- the ABI runs the second one again
    (its own flag is down) and hands back the same result as a compile on its own."""
    import threading
    import time
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock400_1.clq"))
    W = 30000
    mdds = [ddo_amd.Mdd(model, W) for _ in range(2)]
    lb = -(1 << 62)
    t0 = time.perf_counter()
    alone = mdds[1].compile(CompilationType.Relaxed, W, model.root(), lb)
    whole = time.perf_counter() - t0
    assert alone is not None and whole > 0.02
    flags = [C.c_int(0), C.c_int(0)]
    t = threading.Timer(whole / 10, lambda: setattr(flags[0], "value", 1))
    t.start()
    out = ddo_amd.Mdd.compile_batch(mdds, [CompilationType.Relaxed] * 2, [W, W], [model.root(), model.root()], [lb, lb], cutoffs=flags)
    t.join()
    assert out[0] is None, "the first compile ran to the end although its flag went up a tenth into it"
    assert out[1] is not None and out[1].best_value == alone.best_value and out[1].is_exact == alone.is_exact
    assert len(mdds[1].drain_cutset()) > 0


def test_cutoffs_of_concurrent_callers_stay_their_own(oracle):
    """The combining layer under ddo_mdd_compile (Engine::compile_combined): 24 host threads, one mdd each, loop plain compile() on
    wide sub-problems and so share launches.  A third of them carry a Cutoff flag that goes up while they compile: THEIR compile
    comes back as Err(CutoffOccurred) (None) -- the others, cut on the device by the same launch-wide flag, are compiled again by the
    layer and every one of their results equals the oracle's for its input (Cutoff::must_stop is per compile, clean.rs:352)."""
    import threading
    import time

    from tests.parity_util import canon_from_mdd, diff
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock400_1.clq"))
    inst = oracle.misp(data_path("misp", "brock400_1.clq"))
    W = 3000
    _, recs = inst.trace_solve(W, 8)
    recs = [r for r in recs if r["nodes_expanded"] > 50000][:6]
    assert len(recs) >= 4
    T = 24
    errors, cut_seen, done = [], [0], [0]
    flags = [C.c_int(0) if t % 3 == 0 else None for t in range(T)]
    go = threading.Event()

    def worker(t):
        try:
            mdd = ddo_amd.Mdd(model, W)
            go.wait()
            for rep in range(3):
                r = recs[(t + rep) % len(recs)]
                sub = ddo_amd.SubProblem(state=np.array(r["state"], dtype=np.uint64), value=r["value"], path=[], depth=r["depth"])
                comp = mdd.compile(r["comp_type"], r["width"], sub, r["best_lb"], cutoff=flags[t])
                if comp is None:
                    assert flags[t] is not None and flags[t].value == 1, f"thread {t}: cut although its own flag is down"
                    cut_seen[0] += 1
                    return
                d = diff(r, canon_from_mdd(mdd, comp, model.ws))
                if d is not None:
                    errors.append(f"thread {t} rep {rep}: {d}")
                    return
                done[0] += 1
        except Exception as e:
            errors.append(f"thread {t}: {e!r}")

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    go.set()
    time.sleep(0.01)
    for f in flags:
        if f is not None:
            f.value = 1
    for th in ths:
        th.join()
    assert not errors, errors[:3]
    assert done[0] >= 3 * (T - T // 3) and cut_seen[0] + done[0] >= T


def test_best_exact_solution_of_a_relaxed_dd(brock):
    """best_exact_value / best_exact_solution (mdd.rs:96-110): the best terminal reached by an exact path"""
    mdd = ddo_amd.Mdd(brock, 50)
    c = mdd.compile(CompilationType.Relaxed, 50, brock.root(), -(1 << 62))
    assert c is not None and not c.is_exact
    v = mdd.best_exact_value()
    sol = mdd.best_exact_solution()
    if v is None:
        assert sol is None
    else:
        chosen = [d.variable for d in sol if d.value == 1]
        rows, _w = brock.export()
        assert len(chosen) == v and v <= 12 and is_independent_set(rows, brock.ws, chosen)
    # a restricted DD only has exact paths: both views coincide
    mdd.compile(CompilationType.Restricted, 50, brock.root(), -(1 << 62))
    assert mdd.best_exact_value() == mdd.best_value()
    assert [(d.variable, d.value) for d in mdd.best_exact_solution()] == [(d.variable, d.value) for d in mdd.best_solution()]


def test_best_exact_solution_of_a_relaxed_knapsack_dd():
    """a relaxed knapsack DD keeps several terminal nodes, some reached by exact paths only"""
    rng = np.random.default_rng(7)
    profit = rng.integers(1, 100, size=10)
    weight = rng.integers(1, 50, size=10)
    cap = int(weight.sum() // 2)
    model = ddo_amd.Knapsack.from_items(cap, profit, weight)
    mdd = ddo_amd.Mdd(model, 8)
    seen_exact = False
    for w in (2, 4, 6):   # width 6: inexact DD, best exact terminal 428 (same figures from the CPU emulation)
        c = mdd.compile(CompilationType.Relaxed, w, model.root(), -(1 << 62))
        assert c is not None and not c.is_exact
        v, sol = mdd.best_exact_value(), mdd.best_exact_solution()
        assert (v is None) == (sol is None)
        if v is not None:
            seen_exact = True
            taken = [d.variable for d in sol if d.value == 1]
            assert int(profit[taken].sum()) == v and int(weight[taken].sum()) <= cap and len(sol) == 10
            assert v <= mdd.best_value()
    assert seen_exact and v == 428 and mdd.best_value() == 473


@pytest.mark.parametrize("fringe", ["nodup", "lazy"])
def test_set_primal_seeds_the_incumbent(brock, fringe):
    """Solver::set_primal (parallel.rs:630-636): a better primal replaces the incumbent, a worse one is ignored; the
    search seeded with an optimal solution proves it with less work"""
    ref = ParallelSolver(brock, FixedWidth(100), nb_threads=64, fringe=fringe)
    assert ref.maximize().best_value == 12
    opt = ref.best_solution()
    seeded = ParallelSolver(brock, FixedWidth(100), nb_threads=64, fringe=fringe)
    seeded.set_primal(12, opt)
    assert seeded.best_lower_bound() == 12 and seeded.best_value() == 12
    seeded.set_primal(5, opt[:5])            # not better: ignored
    assert seeded.best_lower_bound() == 12
    c = seeded.maximize()
    assert c.is_exact and c.best_value == 12
    assert sorted((d.variable, d.value) for d in seeded.best_solution()) == sorted((d.variable, d.value) for d in opt)
    assert seeded.explored() <= ref.explored()
    assert seeded.counters()["nodes_expanded"] <= ref.counters()["nodes_expanded"]


@pytest.mark.parametrize("fringe", ["nodup", "lazy"])
def test_open_bound_and_device_time_queries(brock, fringe):
    s = ParallelSolver(brock, FixedWidth(100), nb_threads=16, fringe=fringe)
    assert s.step() == 1                      # the root
    s.flush()
    assert s.fringe_len() > 0
    ub0 = s.fringe_best_ub()
    assert 12 <= ub0 <= 200
    for _ in range(3):
        if s.step() != 1:
            break
    s.flush()
    assert s.fringe_best_ub() <= ub0          # best-first: the open bound never increases
    ms, launches = s.device_time()
    assert launches >= 2 and ms > 0.0
    c = s.maximize()
    assert c.is_exact and c.best_value == 12 and s.fringe_len() == 0
    assert s.best_upper_bound() == s.best_lower_bound() == 12 and s.gap() == 0.0


def test_concurrent_compiles_on_distinct_mdds(brock, oracle):
    """include/ddo_hip.h: distinct ddo_mdd objects may be used concurrently (one DecisionDiagram per worker thread,
    parallel.rs:576-602).  All mdds of a (model, device, width) share one device engine: 8 host threads hammer it with
    different sub-problems and every result must equal the oracle's for THAT thread's input."""
    import threading

    from tests.parity_util import canon_from_mdd, diff

    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    _, recs = inst.trace_solve(60, 48)
    assert len(recs) >= 16
    nthreads = 8
    errors = []

    def worker(t):
        try:
            mdd = ddo_amd.Mdd(brock, 60)
            for rep in range(3):
                for r in recs[t::nthreads]:
                    sub = ddo_amd.SubProblem(state=np.array(r["state"], dtype=np.uint64), value=r["value"], path=[], depth=r["depth"])
                    comp = mdd.compile(r["comp_type"], r["width"], sub, r["best_lb"])
                    d = diff(r, canon_from_mdd(mdd, comp, brock.ws))
                    if d is not None:
                        errors.append(f"thread {t} rep {rep}: {d}")
                        return
        except Exception as e:   # e.g. 'a batch is already in flight'
            errors.append(f"thread {t}: {e!r}")

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors[:3]


def test_compile_batch_survives_a_full_output_arena(brock, oracle, monkeypatch):
    """The compiles of one launch share the output arena.  With a 16 MB arena a batch of 128 relaxed compiles at width 3001 (about 40 MB of cut-sets)
    overflows it: ddo_mdd_compile_batch compiles the ones that found it full again on their own (growing the arena when a
    single cut-set does not fit) -- every result must equal the oracle's, as without the squeeze."""
    from tests.parity_util import canon_from_mdd, diff

    monkeypatch.setenv("DDO_HIP_ARENA_MB", "16")
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    W = 3001   # (a width no other test uses: the engine of this (model, width) is created under the small arena)
    _, recs = inst.trace_solve(W, 2)
    relaxed = [r for r in recs if r["comp_type"] == CompilationType.Relaxed and len(r["cutset"]) > 500]
    assert relaxed
    r = relaxed[0]
    B = 128
    mdds = [ddo_amd.Mdd(brock, W) for _ in range(B)]
    sub = ddo_amd.SubProblem(state=np.array(r["state"], dtype=np.uint64), value=r["value"], path=[], depth=r["depth"])
    comps = ddo_amd.Mdd.compile_batch(mdds, [r["comp_type"]] * B, [r["width"]] * B, [sub] * B, [r["best_lb"]] * B)
    for j in range(B):
        d = diff(r, canon_from_mdd(mdds[j], comps[j], brock.ws))
        assert d is None, f"compile #{j}: {d}"


def test_two_callers_with_several_requests_each_share_overflowing_launches(brock, oracle, monkeypatch):
    """a caller of ddo_mdd_compile_batch holds SEVERAL requests of the combining layer.  Under a 256 KB output arena
    some of them finish while others find the arena full and run again: a caller must decode (and release the buffer set of) every
    result the moment it is delivered, or the leader that re-runs its other requests waits for it for ever.  Two threads, 24
    relaxed compiles each with cut-sets of some 30 KB: several rounds of overflow and growth; every result equals the oracle's."""
    import threading
    from tests.parity_util import canon_from_mdd, diff

    monkeypatch.setenv("DDO_HIP_ARENA_KB", "256")
    inst = oracle.misp(data_path("misp", "brock200_2.clq"))
    W = 1503   # (a width no other test uses: its engine is created under the small arena)
    _, recs = inst.trace_solve(W, 3)
    relaxed = [r for r in recs if r["comp_type"] == CompilationType.Relaxed and len(r["cutset"]) > 200]
    assert relaxed
    B = 24
    errors = []

    def caller(t):
        try:
            mdds = [ddo_amd.Mdd(brock, W) for _ in range(B)]
            for rep in range(3):
                rs = [relaxed[(t + j + rep) % len(relaxed)] for j in range(B)]
                subs = [ddo_amd.SubProblem(state=np.array(r["state"], dtype=np.uint64), value=r["value"], path=[], depth=r["depth"]) for r in rs]
                comps = ddo_amd.Mdd.compile_batch(mdds, [r["comp_type"] for r in rs], [r["width"] for r in rs], subs, [r["best_lb"] for r in rs])
                for j in range(B):
                    d = diff(rs[j], canon_from_mdd(mdds[j], comps[j], brock.ws))
                    if d is not None:
                        errors.append(f"caller {t} rep {rep} compile {j}: {d}")
                        return
        except Exception as e:
            errors.append(f"caller {t}: {e!r}")

    ths = [threading.Thread(target=caller, args=(t,), daemon=True) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=240)
    assert not any(th.is_alive() for th in ths), "callers of compile_batch are stuck in the combining layer"
    assert not errors, errors[:3]


def test_a_workspace_capacity_error_is_not_mistaken_for_a_full_arena(monkeypatch):
    """only the shared output arena is worth a second run.  A Pooled decision diagram that outgrows its pool is a normal,
    documented outcome (status <= -100, not the arena): it must come back to its caller at once as a capacity error -- not be run
    again and again while the pinned arenas grow to 8 GB -- and the compile next to it in the same launch must be unharmed."""
    import time
    monkeypatch.setenv("DDO_HIP_POOLED_NODES", "600")
    model = ddo_amd.Misp.read_instance(data_path("misp", "brock200_2.clq"))
    W = 77   # (a width no other test uses: the engine of this (model, width) is created under the small pool)
    mdds = [ddo_amd.Pooled(model, W), ddo_amd.Pooled(model, W)]
    root = model.root()
    tiny = ddo_amd.SubProblem(state=np.array([0x3FF] + [0] * (model.ws - 1), dtype=np.uint64), value=0, path=[], depth=0)
    t0 = time.time()
    with pytest.raises(ddo_amd.DdoError) as ei:
        ddo_amd.Mdd.compile_batch(mdds, [CompilationType.Relaxed] * 2, [W] * 2, [root, tiny], [-(1 << 40)] * 2)
    assert time.time() - t0 < 20, "a per-slot capacity error was re-run as if the arena had overflowed"
    assert "capacity" in str(ei.value).lower(), str(ei.value)
    # the engine is still at full batch size and healthy: a compile that fits runs
    comp = mdds[1].compile(CompilationType.Relaxed, W, tiny, -(1 << 40))
    assert comp.is_exact


_LPT_SCRIPT = r"""
import json, sys
import ddo_amd
from ddo_amd import FixedWidth, ParallelSolver
m = ddo_amd.Misp.read_instance(sys.argv[1])
s = ParallelSolver(m, FixedWidth(int(sys.argv[2])), nb_threads=int(sys.argv[3]), fringe="lazy")
c = s.maximize()
k = s.counters()
print(json.dumps({"best": c.best_value, "exact": c.is_exact, "explored": s.explored(), "nodes": k["nodes_expanded"], "compiles": k["compiles"],
                  "launches": s.device_time()[1], "sol": sorted((d.variable, d.value) for d in s.best_solution())}))
"""


@pytest.mark.parametrize("inst,width,threads", [("brock200_4", 200, 8192), ("johnson8-4-4", 50, 4096)])
def test_the_launch_order_changes_nothing_but_the_order(inst, width, threads, tmp_path):
    """Engine::launch draws the DDs of a launch longest first when there are more of them than node slots (lpt_count_kernel /
    lpt_order_kernel, `P.order`); DDO_HIP_LPT=0 keeps the input order.  Results are matched to inputs by position either way, so
    the whole search -- optimum, explored sub-problems, compiles -- must come out the same."""
    import json
    import os
    import subprocess
    import sys
    script = tmp_path / "lpt_run.py"
    script.write_text(_LPT_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for lpt in ("0", "1"):
        env = dict(os.environ, DDO_HIP_LPT=lpt, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, str(script), data_path("misp", inst + ".clq"), str(width), str(threads)], env=env, cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[lpt] = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = out["0"], out["1"]
    assert a["exact"] and b["exact"] and a["best"] == b["best"], (a, b)
    model = ddo_amd.Misp.read_instance(data_path("misp", inst + ".clq"))
    rows, wd = model.export()
    for o in (a, b):
        taken = [v for v, d in o["sol"] if d == 1]
        assert is_independent_set(rows, model.ws, taken) and int(sum(wd[v] for v in taken)) == o["best"]
    # the search itself is the same search (counters differ by some 1e-6 from run to run in either order: merges are folded by
    # atomics, and two runs of one order are not bit-identical either)
    for k in ("explored", "compiles", "nodes"):
        assert abs(a[k] - b[k]) <= 0.02 * a[k], (k, a, b)
