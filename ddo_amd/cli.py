"""Command-line front ends with the arguments and the report of the reference's example binaries, running the B&B on
the GPU engine:

    python -m ddo_amd.cli misp     <file> [-t N] [-d SECONDS] [-w WIDTH]      (examples/misp/main.rs:223-236, 340-397)
    python -m ddo_amd.cli knapsack <file> [-t N] [-d SECONDS] [-w WIDTH]      (examples/knapsack/main.rs:230-246, 308-358)
    python -m ddo_amd.cli max2sat  -f <file> [-w WIDTH] [-t SECONDS]          (examples/max2sat/main.rs:19-72)
    python -m ddo_amd.cli mcp      -f <file> [-w WIDTH] [-t SECONDS]          (examples/mcp/main.rs:19-66)
    python -m ddo_amd.cli tsptw    <file> [-w FACTOR] [-t N] [-d SECONDS]     (examples/tsptw/main.rs:49-101)

`-t/--threads` of misp / knapsack is the number of sub-problems compiled concurrently (the reference's worker threads);
on the GPU the useful values are hundreds to thousands (tools/fringe_compare.py), so the default is 8192 for misp (brock400_1 / W = 10 000 is proved in 183 / 169 / 146 / 143 / 138 s with
1024 / 2048 / 4096 / 8192 / 16384 in flight) and 256 for knapsack instead of 8.  The report has the reference's lines
(`Duration / Objective / Upper Bnd / Lower Bnd / Gap / Aborted / [Cost] / Solution`) so that the outputs can be diffed.
Like the reference's binaries the solvers use the duplicate-free fringe; `--fringe lazy` selects the device-resident
block fringe (MISP only).  The knapsack binary of the reference couples a frontier cut-set, a cache and a dominance checker
(SeqCachingSolverFc + SimpleDominanceChecker(KPDominance)): so does this one."""
import argparse
import re
import sys
import time

from . import binding as B


def _width(model, w):  # `max_width`, e.g. misp/main.rs:322-328
    return B.FixedWidth(w) if w is not None else B.NbUnassignedWidth(model.nb_variables())


def _cutoff(seconds):  # `cutoff`, e.g. misp/main.rs:331-337
    return B.TimeBudget(seconds) if seconds else B.NoCutoff()


def _rust_bool(b):
    return "true" if b else "false"


def _report(duration, completion, solver, solution_text, extra=()):
    ub, lb = solver.best_upper_bound(), solver.best_lower_bound()
    i64max, i64min = (1 << 63) - 1, -(1 << 63)
    print(f"Duration:   {duration:.3f} seconds")
    print(f"Objective:  {completion.best_value if completion.best_value is not None else -1}")
    print(f"Upper Bnd:  {min(ub, i64max)}")
    print(f"Lower Bnd:  {max(lb, i64min)}")
    print(f"Gap:        {solver.gap():.3f}")
    print(f"Aborted:    {_rust_bool(not completion.is_exact)}")
    for line in extra:
        print(line)
    print(f"Solution:   {solution_text}")


def _sorted_solution(solver):
    sol = solver.best_solution()
    if sol is None:
        return None
    return sorted(sol, key=lambda d: d.variable)


def _list(xs):
    return "[" + ", ".join(str(x) for x in xs) + "]"


def _solve(model, args, threads, seconds, fringe="nodup", **solver_kw):
    solver = B.ParallelSolver(model, _width(model, args.width), _cutoff(seconds), nb_threads=threads, device=args.device,
                              fringe=fringe, **solver_kw)
    t0 = time.perf_counter()
    completion = solver.maximize()
    return solver, completion, time.perf_counter() - t0


def misp(args):
    model = B.Misp.read_instance(args.fname)
    solver, completion, dt = _solve(model, args, args.threads, args.duration, args.fringe)
    sol = _sorted_solution(solver)
    chosen = [d.variable for d in sol if d.value == 1] if sol is not None else []
    rows, _w = model.export()         # complement-adjacency rows: bit b of row a set <=> a and b may both be chosen
    for i, a in enumerate(chosen):    # the reference's own check of the answer (main.rs:380-388)
        for b in chosen[i + 1:]:
            if not (int(rows[a * model.ws + b // 64]) >> (b % 64)) & 1:
                print(f"not a solution ! {a} -- {b}")
    _report(dt, completion, solver, _list(chosen))


def knapsack(args):
    model = B.Knapsack.read_instance(args.fname)
    # The reference's knapsack binary (main.rs:320-337) is SeqCachingSolverFc: frontier cut-set + SimpleCache, with
    # SimpleDominanceChecker(KPDominance) -- the same three pieces here, on the device.  It parses --duration (default 30)
    # but builds its solver with NoCutoff (main.rs:326); the budget is honoured here so that a hard instance at the
    # default width of 2 ends with a gap instead of running for hours.
    solver, completion, dt = _solve(model, args, args.threads, args.duration, cutset_type=B.FRONTIER, cache_entries=1 << 20,
                                    dominance_entries=1 << 14)
    sol = _sorted_solution(solver)
    _report(dt, completion, solver, _list([d.value for d in sol] if sol is not None else []))


def _read_wcnf(path):
    """(nb_vars, {(a, b): w}) as examples/max2sat/data.rs:65-112 reads it: a later clause on the same pair of literals
    replaces the earlier one."""
    pb = re.compile(r"^p\s+wcnf\s+(\d+)\s+(\d+)")
    binary = re.compile(r"^(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+0")
    unit = re.compile(r"^(-?\d+)\s+(-?\d+)-?\s+0")
    n, weights = 0, {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or re.match(r"^c\s.*$", line):
                continue
            m = pb.match(line)
            if m:
                n = int(m.group(1))
                continue
            m = binary.match(line)
            if m:
                w, x, y = (int(g) for g in m.groups())
                weights[(min(x, y), max(x, y))] = w
                continue
            m = unit.match(line)
            if m:
                w, x = int(m.group(1)), int(m.group(2))
                weights[(x, x)] = w
    return n, weights


def _max2sat_cost(path, sol):
    """weight of the clauses the assignment falsifies (`solution_cost`, max2sat/main.rs:89-118)"""
    if sol is None:
        return 0
    _n, weights = _read_wcnf(path)
    value = {d.variable: d.value for d in sol}

    def false_lit(lit):
        return value.get(abs(lit) - 1, 0) == (-1 if lit > 0 else 1)

    return sum(w for (a, b), w in weights.items() if false_lit(a) and false_lit(b))


def max2sat(args):
    model = B.Max2Sat.read_instance(args.file)
    solver, completion, dt = _solve(model, args, args.concurrent, args.timeout)
    sol = _sorted_solution(solver)
    lits = [(1 + d.variable) * d.value for d in sol] if sol is not None else []
    _report(dt, completion, solver, _list(lits), extra=[f"Cost:       {_max2sat_cost(args.file, sol)}"])


def mcp(args):
    model = B.Mcp.read_instance(args.file)
    solver, completion, dt = _solve(model, args, args.concurrent, args.timeout)
    sol = solver.best_solution() or []   # printed as the solver returns it (mcp/main.rs:56, 65)
    _report(dt, completion, solver, _list(f"Decision {{ variable: Variable({d.variable}), value: {d.value} }}" for d in sol))


def tsptw(args):
    """examples/tsptw/main.rs:70-128: DefaultCachingSolver (frontier cut-set + SimpleCache), SimpleDominanceChecker(TsptwDominance),
    TsptwWidth(nb_vars, -w); the six report lines of print_solution"""
    import os
    import numpy as np
    model = B.Tsptw.read_instance(args.instance)
    solver = B.ParallelSolver(model, B.TsptwWidth(args.width or 1), _cutoff(args.duration), nb_threads=args.threads, device=args.device,
                              fringe="nodup", cutset_type=B.FRONTIER, cache_entries=1 << 22, dominance_entries=1 << 22)
    t0 = time.perf_counter()
    completion = solver.maximize()
    dt = time.perf_counter() - t0

    def objective(x):   # main.rs:109-115
        if x <= -(1 << 63):
            return "+inf"
        if x >= (1 << 63) - 1:
            return "-inf"
        return f"{-(np.float32(x) / np.float32(10000.0)):.2f}"

    sol = _sorted_solution(solver)
    path = os.path.abspath(args.instance)
    print(f"instance : {os.path.basename(os.path.dirname(path))}/{os.path.basename(path)}")
    print(f"status   : {'Proved' if completion.is_exact else 'Timeout'}")
    print(f"lower bnd: {objective(solver.best_lower_bound())}")
    print(f"upper bnd: {objective(solver.best_upper_bound())}")
    print(f"duration : {dt}")
    print("solution : " + ("No feasible solution found" if sol is None else "".join(f" {d.value}" for d in sol)))   # main.rs:130-146: every entry preceded by a space


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ddo_amd.cli", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name, fn in (("misp", misp), ("knapsack", knapsack)):
        p = sub.add_parser(name)
        p.add_argument("fname", help="the path to the instance file")
        p.add_argument("-t", "--threads", type=int, default=8192 if name == "misp" else 256,
                       help="sub-problems compiled concurrently")
        p.add_argument("-d", "--duration", type=int, default=None if name == "misp" else 30,
                       help="the maximum amount of time (s) the solver may run")
        p.add_argument("-w", "--width", type=int, default=None, help="the maximum number of nodes per layer")
        p.add_argument("--fringe", choices=("nodup", "lazy"), default="nodup")
        p.add_argument("--device", type=int, default=0)
        p.set_defaults(fn=fn)
    p = sub.add_parser("tsptw")
    p.add_argument("instance", help="the path to the TSP+TW instance")
    p.add_argument("-w", "--width", type=int, default=None, help="multiplier of the default width nb_vars * (depth + 1)")
    p.add_argument("-t", "--threads", type=int, default=64, help="sub-problems compiled concurrently")
    p.add_argument("-d", "--duration", type=int, default=None, help="time budget in seconds")
    p.add_argument("--device", type=int, default=0)
    p.set_defaults(fn=tsptw)
    for name, fn in (("max2sat", max2sat), ("mcp", mcp)):
        p = sub.add_parser(name)
        p.add_argument("-f", "--file", required=True, help="the instance file")
        p.add_argument("-w", "--width", type=int, default=None, help="maximum width in a layer")
        p.add_argument("-t", "--timeout", type=int, default=None, help="max time to find the solution")
        p.add_argument("--concurrent", type=int, default=256, help="sub-problems compiled concurrently")
        p.add_argument("--device", type=int, default=0)
        p.set_defaults(fn=fn)
    args = ap.parse_args(argv)
    if args.cmd == "knapsack" and args.width is None:
        args.width = 2   # FixedWidth(2) when no width is given (knapsack/main.rs:320-324)
    try:
        args.fn(args)
    except B.DdoError as e:
        print(f"error: {e}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
