"""Diagnosis: the pooled + SimpleCache replay on the GPU against the host emulation of the same device source, compile by compile:
entries in use in the two cache tables after every compile, and the first compile whose record differs from the oracle's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np   # noqa: E402

import ddo_amd   # noqa: E402
from ddo_amd import SubProblem   # noqa: E402
from tests.dd_wire import IN_CACHE, IN_WANT_PATHS   # noqa: E402
from tests.emul_binding import Emul   # noqa: E402
from tests.oracle_binding import Oracle   # noqa: E402
from tests.parity_util import canon_from_mdd, diff   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, width, maxc = (sys.argv[1:4] + ["johnson8-4-4", "4", "300"][len(sys.argv) - 1:])[:3]
path = os.path.join(ROOT, "data", "misp", name + ".clq")
o = Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
inst = o.misp(path)
_, recs = o.trace_ex("misp+pooled", path, int(width), int(maxc), False, True)
model = ddo_amd.Misp.read_instance(path)
e = Emul(inst.n, inst.rows, inst.weights, 14000, engine=2)
e.pooled(True)
e.pooled_cache(1 << 16)
cache = ddo_amd.SimpleCache(model, 1 << 16)
mdd = ddo_amd.Pooled(model, max(max(int(r["width"]) for r in recs), 8), caching=True)
for i, r in enumerate(recs):
    g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=IN_WANT_PATHS | IN_CACHE)[0]
    sub = SubProblem(state=np.array(r["state"], dtype=np.uint64), value=r["value"], path=[], depth=r["depth"])
    comp = mdd.compile(r["comp_type"], r["width"], sub, r["best_lb"], cache=cache)
    dg = diff(r, canon_from_mdd(mdd, comp, model.ws))
    de = diff(r, g)
    print(f"#{i} type={r['comp_type']} depth={r['depth']} nodes={r['nodes_expanded']} exact={r['is_exact']} | emul used {e.cache_used()} hits {g['cache_hits']} {'ok' if de is None else de} | "
          f"gpu used {cache.stats()['used']} {'ok' if dg is None else dg}", flush=True)
    if dg is not None or de is not None or e.cache_used() != cache.stats()["used"]:
        break
