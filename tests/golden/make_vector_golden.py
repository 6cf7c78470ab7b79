"""Generates tests/golden/vector_compile_golden.json from the CPU oracle: compiles of the signed-vector models
(MAX2SAT, MCP), taken from traced sequential searches (oracle_{max2sat,mcp}_trace_solve).

The reference (Rust) cannot run in this image; these vectors are oracle outputs, the oracle being pinned on the
reference's known optima for these models (examples/max2sat/tests.rs:65-105, examples/mcp/tests.rs:64-103).  Ties of the
ranking are broken by the packed state words in oracle and device alike (oracle/models.hpp compare_signed_vectors), ties
between equal-valued best arcs in favour of an exact best path (Problem::canonical_ties).  Each case is one compile():
inputs + everything observable; the cut-set as (count, digest).  Run from the repo root:
    python tests/golden/make_vector_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.oracle_binding import Oracle  # noqa: E402
from tests.parity_util import cutset_digest  # noqa: E402

CASES = [   # (kind, file, width, compiles kept)
    ("max2sat", "frb10-6-1.wcnf", 5000, 6),     # BASELINE config C3
    ("max2sat", "frb10-6-1.wcnf", 100, 12),
    ("max2sat", "frb10-6-3.wcnf", 0, 10),       # NbUnassignedWidth, as the reference's own test
    ("max2sat", "pass.wcnf", 2, 8),
    ("max2sat", "negative_wt.wcnf", 2, 6),
    ("mcp", "mcp_n30_p0.1_002.mcp", 50, 12),
    ("mcp", "mcp_n30_p0.1_005.mcp", 5, 16),
    ("mcp", "mcp_n30_p0.1_009.mcp", 0, 10),
]


def main():
    o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    cases = []
    for kind, fname, width, keep in CASES:
        _, recs = o.vector_trace(kind, os.path.join(ROOT, "data", kind, fname), width, keep)
        for i, r in enumerate(recs[:keep]):
            cases.append({
                "id": f"{kind}-{fname.split('.')[0]}-w{width}-#{i}", "kind": kind, "file": fname, "comp_type": r["comp_type"],
                "width": int(r["width"]), "best_lb": int(r["best_lb"]), "state": [str(int(x)) for x in r["state"]],
                "value": int(r["value"]), "depth": int(r["depth"]), "is_exact": r["is_exact"], "best_value": r["best_value"],
                "best_exact_value": r["best_exact_value"], "nodes_expanded": int(r["nodes_expanded"]), "arcs": int(r["arcs"]),
                "layers": int(r["layers"]), "n_cutset": len(r["cutset"]), "cutset_digest": cutset_digest(r["cutset"]),
            })
    out = os.path.join(ROOT, "tests", "golden", "vector_compile_golden.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_vector_golden.py", "source": "CPU oracle (oracle/ddo_oracle.hpp, oracle/models.hpp)",
                   "cases": cases}, f, indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
