"""Generates tests/golden/misp_pooled_golden.json from the CPU oracle's Pooled<S> (oracle/ddo_oracle.hpp, restating
mdd/pooled.rs:117-823 and pinned on pooled.rs's unit tests, oracle/kat_main.cpp): single compiles of MISP sub-problems as
POOLED decision diagrams -- root sub-problems and the first cut-set nodes of the relaxed root DD, restricted and relaxed, two
widths.  The fixture pins the oracle against regressions and is the target a device variant of the pooled DD has to hit
(counters, values, (count, digest) of the frontier cut-set, best path).  The reference itself (Rust) cannot be run here.
Run from the repo root:  python tests/golden/make_pooled_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.oracle_binding import Oracle  # noqa: E402
from tests.parity_util import cutset_digest  # noqa: E402

INSTANCES = ["johnson8-4-4", "c-fat200-1", "MANN_a9", "hamming6-4", "brock200_2"]
WIDTHS = [5, 50]
LOWEST = -(1 << 40)


def case_of(inst, name, tag, comp_type, width, lb, state, value, depth):
    r = inst.compile(comp_type, width, lb, state, value, depth, pooled=True)
    return {"id": f"{name}-{tag}-{'restricted' if comp_type == 2 else 'relaxed'}-w{width}", "instance": name, "comp_type": comp_type, "width": width,
            "best_lb": lb, "state": [str(int(x)) for x in state], "value": value, "depth": depth, "is_exact": r["is_exact"],
            "best_value": r["best_value"], "best_exact_value": r["best_exact_value"], "nodes_expanded": r["nodes_expanded"], "arcs": r["arcs"],
            "layers": r["layers"], "n_cutset": len(r["cutset"]), "cutset_digest": cutset_digest(r["cutset"]),
            "best_path": [list(p) for p in r["best_path"]]}, r


def main():
    o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    cases = []
    for name in INSTANCES:
        inst = o.misp(os.path.join(ROOT, "data", "misp", name + ".clq"))
        root = inst.root_state()
        for w in WIDTHS:
            c, rr = case_of(inst, name, "root", 2, w, LOWEST, root, 0, 0)
            cases.append(c)
            lb = rr["best_value"] if rr["best_value"] is not None else LOWEST
            c, rx = case_of(inst, name, "root", 1, w, lb, root, 0, 0)
            cases.append(c)
            for k, (state, value, ub, depth) in enumerate(rx["cutset"][:2]):   # sub-problems below the root
                import numpy as np
                st = np.array([int(x) for x in state], dtype=np.uint64)
                for ct in (2, 1):
                    c, _ = case_of(inst, name, f"cut{k}", ct, w, lb, st, int(value), int(depth))
                    cases.append(c)
    out = os.path.join(ROOT, "tests", "golden", "misp_pooled_golden.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_pooled_golden.py", "source": "CPU oracle Pooled<S> (oracle/ddo_oracle.hpp; pooled.rs:117-823)",
                   "cases": cases}, f, indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
