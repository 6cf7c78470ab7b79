"""Generates tests/golden/misp_wide_golden.json from the CPU oracle: brock400_1's root decision diagrams at width 100 000
(SURVEY.md section 8 d2's micro grid asks for W in {1k, 10k, 100k}; the oracle needs about six minutes for the two compiles,
too long for a test, so their observable outputs are kept as a fixture -- counters plus (count, digest) of the cut-set).
Run from the repo root:  python tests/golden/make_wide_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.oracle_binding import Oracle  # noqa: E402
from tests.parity_util import cutset_digest  # noqa: E402


def main():
    o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    inst = o.misp(os.path.join(ROOT, "data", "misp", "brock400_1.clq"))
    root = inst.root_state()
    lowest = -(1 << 40)
    W = 100000
    cases = []
    rr = inst.compile(2, W, lowest, root, 0, 0)
    for comp_type, lb, tag in ((2, lowest, "restricted"), (1, rr["best_value"], "relaxed")):
        r = rr if comp_type == 2 else inst.compile(1, W, lb, root, 0, 0)
        cases.append({"id": f"brock400_1-{tag}-w{W}", "instance": "brock400_1", "comp_type": comp_type, "width": W, "best_lb": lb,
                      "state": [str(int(x)) for x in root], "value": 0, "depth": 0, "is_exact": r["is_exact"], "best_value": r["best_value"],
                      "best_exact_value": r["best_exact_value"], "nodes_expanded": r["nodes_expanded"], "arcs": r["arcs"], "layers": r["layers"],
                      "n_cutset": len(r["cutset"]), "cutset_digest": cutset_digest(r["cutset"])})
    out = os.path.join(ROOT, "tests", "golden", "misp_wide_golden.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_wide_golden.py", "source": "CPU oracle (oracle/ddo_oracle.hpp)", "cases": cases}, f, indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
