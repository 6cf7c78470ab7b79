"""CPU suite: the TSPTW model (examples/tsptw: BASELINE config C5) of the layer-rebuilding device engine -- variable fan-out (one
child per node that may be visited next), fuzzy relaxed states, TsptwDominance, frontier cut-set and SimpleCache -- compiled
as the lock-step host emulation and replayed against traced sequential searches of the oracle, compile by compile, in order
(cache and dominance checker are shared by all compiles of a search: the replay is stateful)."""
import pytest

import ddo_amd
from tests.conftest import data_path
from tests.dd_wire import IN_CACHE, IN_DOMINANCE, IN_FRONTIER, IN_MUST_EXPLORE, IN_WANT_PATHS
from tests.emul_binding import ModelEmul
from tests.parity_util import diff

CONFIGS = [("tsptw", False, False), ("tsptw", True, False), ("tsptw", True, True), ("tsptw+dominance", False, False), ("tsptw+dominance", True, True)]


@pytest.mark.parametrize("kind,frontier,cache", CONFIGS, ids=["lel", "frontier", "frontier+cache", "lel+dominance", "frontier+cache+dominance"])
@pytest.mark.parametrize("fname,width,max_compiles", [("N20ft301", 2, 0), ("N20ft405", 3, 0), ("N20ft308", 1, 40), ("N40ft403", 3, 80), ("N40ft207", 2, 60),
                                                      ("N60ft204", 2, 50), ("N40ft201", 0, 0), ("N20ft402", 4, 60)])
def test_tsptw_replay_of_oracle_search(oracle, fname, width, max_compiles, kind, frontier, cache):
    """width 0 = TsptwWidth(nb_vars, 1) (examples/tsptw/tests.rs:42: the root DD is then exact for these instances); small fixed
    widths force restrictions, merges (TsptwRelax::merge, relax.rs:65-191) and real branch-and-bound"""
    path = data_path("tsptw", "Langevin", fname + ".dat")
    model = ddo_amd.Tsptw.read_instance(path)
    assert model.ws == 5
    summary, recs = oracle.trace_ex(kind, path, width, max_compiles, frontier, cache)
    assert recs
    e = ModelEmul(model, max(int(r["width"]) for r in recs))
    e.keep_layers(True, 1 << 15 if cache else 0)
    if "dominance" in kind:
        e.dominance(1 << 15)
    inexact = 0
    for i, r in enumerate(recs):
        fl = IN_WANT_PATHS | (IN_DOMINANCE if "dominance" in kind else 0) | (IN_FRONTIER if frontier else 0) | (IN_CACHE if cache else 0)
        if cache and r["comp_type"] == 2:
            fl |= IN_MUST_EXPLORE
        g = e.compile(r["comp_type"], r["width"], r["best_lb"], r["state"], r["value"], r["depth"], flags=fl)[0]
        assert g is not None and g["status"] == 0, f"{fname} compile #{i}: status {None if g is None else g['status']}"
        d = diff(r, g)
        assert d is None, f"{fname} W={width} {kind} frontier={frontier} cache={cache} compile #{i} type={r['comp_type']} depth={r['depth']}: {d}"
        inexact += (r["comp_type"] == 1 and not r["is_exact"])
    if width in (2, 3) and fname in ("N20ft405", "N40ft403", "N60ft204"):
        assert inexact > 0


def test_tsptw_model_host_side(tmp_path):
    """instance.rs:52-109 (`(f32 * 10000.0) as usize`), model.rs:36-47 (initial state: at the depot, everything else to visit)"""
    p = tmp_path / "t.dat"
    p.write_text("# tiny\n3\n0 1.5 2\n1.5 0 1.25\n2 1.25 0\n0 100\n1 50.5\n2 60\n")
    m = ddo_amd.Tsptw.read_instance(p)
    assert m.n == 3 and m.ws == 5 and m.initial_value() == 0
    s = m.initial_state()
    assert [int(x) for x in s] == [0, 0b110, 0, 0, 0]
    with pytest.raises(ddo_amd.DdoError):
        ddo_amd.Tsptw.from_arrays([[0] * 70] * 70, [0] * 70, [1] * 70)      # more than 64 nodes
    a, b = s.copy(), s.copy()
    b[4] = 1 << 32                                                           # deeper state ranks higher (TsptwRanking)
    assert m.compare(a, b) < 0 and m.compare(b, a) > 0 and m.compare(a, a) == 0
