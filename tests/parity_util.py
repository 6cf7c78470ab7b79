"""Shared helpers of the parity tests: canonical form of a compile() through the C ABI."""
import hashlib

import numpy as np

import ddo_amd
from ddo_amd import CompilationType, SubProblem


def canon_from_mdd(mdd, completion, ws):
    """Everything observable through `DecisionDiagram` (mdd.rs:75-114), order independent."""
    cut = mdd.drain_cutset()
    return {
        "is_exact": bool(completion.is_exact),
        "best_value": completion.best_value,
        "best_exact_value": mdd.best_exact_value(),
        "cutset": sorted((tuple(int(x) for x in n.state[:ws]), int(n.value), int(n.ub), int(n.depth)) for n in cut),
        "cutset_nodes": cut,
        **{k: v for k, v in mdd.counters().items() if k != "compiles"},
    }


def cutset_digest(cut):
    """Checksum of a canonical cut-set [(state words.., value, ub, depth)] -- a checksum of checksums."""
    h = hashlib.sha256()
    for state, value, ub, depth in cut:
        h.update(np.array(list(state) + [value, ub, depth], dtype=np.int64).tobytes() if False else
                 (",".join(str(x) for x in state) + f"|{value}|{ub}|{depth};").encode())
    return h.hexdigest()[:32]


KEYS = ["is_exact", "best_value", "best_exact_value", "nodes_expanded", "arcs", "layers", "cutset"]


def diff(expected, got, keys=KEYS):
    """First differing key between two canonical records (None when equal)."""
    for k in keys:
        if expected[k] != got[k]:
            if k == "cutset":
                so, sg = set(expected[k]), set(got[k])
                return f"cutset: expected {len(expected[k])} nodes, got {len(got[k])}; only expected {list(so - sg)[:2]}; only got {list(sg - so)[:2]}"
            return f"{k}: expected {expected[k]!r}, got {got[k]!r}"
    return None


def is_independent_set(rows, ws, chosen):
    """`rows` are COMPLEMENT adjacency rows: a,b compatible <=> bit b of row a is set (main.rs:381-388)."""
    for i, a in enumerate(chosen):
        for b in chosen[i + 1:]:
            if not (int(rows[a * ws + b // 64]) >> (b % 64)) & 1:
                return False
    return True


ENGINES = ["auto", "full", "dense", "tier0", "tier1"]   # DDO_MDD_ENGINE_*: the kernels the lazy solver (and bench.py) run; "auto" = no selector
# (ddo_mdd_create picks: at widths of 2048 and more the dense kernel with the full-width engine behind it, else the one engine of the width)
TIER_CAP = {"tier0": 256, "tier1": 1024}


def engine_width(engine, maxw):
    """max_width an mdd bound to `engine` must be created with to serve compiles of width <= maxw (a capacity tier exists
    only under a full-width engine at least twice its layer capacity)."""
    return max(int(maxw), 2 * TIER_CAP.get(engine, 0))


def may_hand_up(engine, rec):
    """A capacity tier hands a DD up when a layer outgrows its node slots or a squash is needed (it has no squash phases);
    a DD whose layers all fit both the tier and the width must complete.  nodes_expanded bounds every layer from above
    (is_exact alone does not tell: a squashed relaxed DD whose best path stayed exact reports is_exact too)."""
    return engine in TIER_CAP and (not rec["is_exact"] or rec["nodes_expanded"] > min(TIER_CAP[engine] // 2, int(rec["width"])))


def replay_records(model, recs, device=0, batch=64, engine="auto"):
    """Replays oracle trace records on the GPU through ddo_mdd_compile_batch; yields (i, record, canonical); canonical is
    ddo_amd.HANDED_UP when the capacity tier the mdds are bound to answered DDO_HANDED_UP."""
    maxw = engine_width(engine, max(int(r["width"]) for r in recs))
    mdds = [ddo_amd.Mdd(model, maxw, device=device, engine=engine) for _ in range(min(batch, len(recs)))]
    for base in range(0, len(recs), batch):
        chunk = recs[base:base + batch]
        ms = mdds[:len(chunk)]
        subs = [SubProblem(state=np.array(r["state"], dtype=np.uint64), value=r["value"], path=[], depth=r["depth"])
                for r in chunk]
        comps = ddo_amd.Mdd.compile_batch(ms, [r["comp_type"] for r in chunk], [r["width"] for r in chunk], subs,
                                          [r["best_lb"] for r in chunk])
        for j, r in enumerate(chunk):
            yield base + j, r, (comps[j] if comps[j] is ddo_amd.HANDED_UP else canon_from_mdd(ms[j], comps[j], model.ws))


def check_replay(model, recs, engine, what, min_completed=1, **kw):
    """Every record replayed on `engine` equals the oracle's; a capacity tier may hand up only what cannot fit it."""
    completed = handed = 0
    for i, r, got in replay_records(model, recs, engine=engine, **kw):
        if got is ddo_amd.HANDED_UP:
            assert may_hand_up(engine, r), f"{what} [{engine}] compile #{i}: handed up although exact with {r['nodes_expanded']} nodes"
            handed += 1
            continue
        d = diff(r, got)
        assert d is None, f"{what} [{engine}] compile #{i} type={r['comp_type']} depth={r['depth']} lb={r['best_lb']}: {d}"
        completed += 1
    assert completed + handed == len(recs) and completed >= min_completed, (what, engine, completed, handed)
    return completed, handed
