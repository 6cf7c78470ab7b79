"""Generates tests/golden/misp_compile_golden.json from the CPU oracle (oracle/_build/liboracle.so).

The reference (Rust) cannot run in this image, so the golden vectors are outputs of the oracle,
which is itself pinned on the reference's known-answer tests (oracle/kat_main.cpp,
tests/test_oracle_kat.py).  Each case is one compile(): inputs + expected observable outputs; the
cut-set is stored as (count, digest).  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.oracle_binding import Oracle  # noqa: E402
from tests.parity_util import cutset_digest  # noqa: E402


def main():
    o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    cases = []

    def add(inst_name, inst, comp_type, width, best_lb, state, value, depth, tag):
        r = inst.compile(comp_type, width, best_lb, state, value, depth)
        cases.append({
            "id": f"{inst_name}-{tag}", "instance": inst_name, "comp_type": comp_type, "width": width,
            "best_lb": best_lb, "state": [str(int(x)) for x in state], "value": value, "depth": depth,
            "is_exact": r["is_exact"], "best_value": r["best_value"], "best_exact_value": r["best_exact_value"],
            "nodes_expanded": r["nodes_expanded"], "arcs": r["arcs"], "layers": r["layers"],
            "n_cutset": len(r["cutset"]), "cutset_digest": cutset_digest(r["cutset"]),
        })

    lowest = -(1 << 40)
    for name, widths in [("brock200_2", [1, 10, 1000]), ("brock400_1", [100, 10000]), ("keller4", [64]),
                         ("p_hat300-1", [300]), ("c-fat500-2", [500]), ("hamming8-4", [256])]:
        inst = o.misp(os.path.join(ROOT, "data", "misp", name + ".clq"))
        root = inst.root_state()
        for w in widths:
            rr = inst.compile(2, w, lowest, root, 0, 0)
            add(name, inst, 2, w, lowest, root, 0, 0, f"restricted-w{w}")
            add(name, inst, 1, w, rr["best_value"], root, 0, 0, f"relaxed-w{w}")
            # a deeper sub-problem: first node of the relaxed cut-set, with a tight lower bound
            rx = inst.compile(1, w, rr["best_value"], root, 0, 0)
            if rx["cutset"]:
                st, val, ub, dep = rx["cutset"][len(rx["cutset"]) // 2]
                import numpy as np
                stw = np.array(list(st), dtype=np.uint64)
                add(name, inst, 2, w, rr["best_value"], stw, val, dep, f"sub-restricted-w{w}")
                add(name, inst, 1, w, rr["best_value"], stw, val, dep, f"sub-relaxed-w{w}")
    out = os.path.join(ROOT, "tests", "golden", "misp_compile_golden.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "source": "CPU oracle (oracle/ddo_oracle.hpp)",
                   "cases": cases}, f, indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
