"""Generates tests/golden/b1_brock400_golden.json: sub-problems of brock400_1's root cut-set at width 10 000 (BASELINE config C4's
instance and width) and, for each, the digests of the two compiles the reference's process_one_node makes of it (parallel.rs:391-437:
restricted, then -- the restricted DD being inexact -- relaxed with the incumbent the restricted one found), computed by the CPU
ORACLE.  tests/test_gpu_boundary_b1.py drives the same sub-problems through plain ddo_mdd_compile from 64 / 512 / 2048 host
threads and compares every compile's digest.  Run from the repo root: python tests/golden/make_b1_golden.py  (about two minutes)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ddo_amd.boundary import cutset_hash   # noqa: E402
from tests.oracle_binding import Oracle   # noqa: E402

WIDTH, NSUB = 10000, 16
CT_RELAXED, CT_RESTRICTED = 1, 2


def digest(rec, with_cutset):
    cut = rec["cutset"] if with_cutset else []
    return {"status": 0, "is_exact": int(rec["is_exact"]), "has_best": int(rec["best_value"] is not None),
            "has_best_exact": int(rec["best_exact_value"] is not None), "best_value": rec["best_value"] or 0,
            "best_exact_value": rec["best_exact_value"] or 0, "nodes_expanded": rec["nodes_expanded"], "arcs": rec["arcs"],
            "layers": rec["layers"], "n_cutset": len(cut), "cutset_hash": cutset_hash(cut)}


def main():
    oracle = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    inst = oracle.misp(os.path.join(ROOT, "data", "misp", "brock400_1.clq"))
    lb0 = -(1 << 40)
    r = inst.compile(CT_RESTRICTED, WIDTH, lb0, inst.root_state(), 0, 0)
    lb = r["best_exact_value"]
    x = inst.compile(CT_RELAXED, WIDTH, lb, inst.root_state(), 0, 0)
    cut = x["cutset"]
    assert not x["is_exact"] and len(cut) > NSUB
    picks = [cut[(len(cut) * k) // NSUB] for k in range(NSUB)]   # sorted by state: evenly spread over the layer
    subs = []
    for state, value, ub, depth in picks:
        a = inst.compile(CT_RESTRICTED, WIDTH, lb, np.array(state, dtype=np.uint64), value, depth)
        lb2 = max(lb, a["best_exact_value"]) if a["best_exact_value"] is not None else lb
        b = None if a["is_exact"] else inst.compile(CT_RELAXED, WIDTH, lb2, np.array(state, dtype=np.uint64), value, depth)
        subs.append({"state": [int(w) for w in state], "value": value, "ub": ub, "depth": depth, "restricted": digest(a, False),
                     "relaxed": None if b is None else digest(b, not b["is_exact"])})
        print(len(subs), a["nodes_expanded"], None if b is None else (b["nodes_expanded"], len(b["cutset"])), flush=True)
    out = {"instance": "brock400_1.clq", "width": WIDTH, "best_lb": lb, "root_cutset": len(cut), "subproblems": subs}
    with open(os.path.join(ROOT, "tests", "golden", "b1_brock400_golden.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))


if __name__ == "__main__":
    main()
