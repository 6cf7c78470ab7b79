"""ctypes mirror of ddo_amd/csrc/dd_types.h (device <-> host wire format) used by the
host-emulation tests.  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

MAX_WS = 72
CT_EXACT, CT_RELAXED, CT_RESTRICTED = 0, 1, 2
IN_FUSED, IN_FILTER_CUTSET, IN_WANT_PATHS = 1, 2, 4
IN_FRONTIER, IN_CACHE, IN_MUST_EXPLORE, IN_MARK_EXPLORED, IN_DOMINANCE, IN_PATH_BITS = 16, 32, 64, 128, 256, 512
ST_OK, ST_CUTOFF, ST_NOT_RUN = 0, 1, 77


class DDInput(C.Structure):
    _fields_ = [("comp_type", C.c_int32), ("flags", C.c_uint32), ("width", C.c_int32), ("value", C.c_int32),
                ("depth", C.c_int32), ("pad", C.c_int32), ("best_lb", C.c_int64), ("src_off", C.c_uint64), ("src_row", C.c_uint32),
                ("pad2", C.c_uint32), ("state", C.c_uint64 * MAX_WS)]


class DDResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("comp_type", C.c_int32), ("is_exact", C.c_int32),
                ("has_exact_best_path", C.c_int32), ("has_best", C.c_int32), ("has_best_exact", C.c_int32),
                ("best_value", C.c_int32), ("best_exact_value", C.c_int32), ("n_layers", C.c_int32), ("lel", C.c_int32),
                ("n_cutset", C.c_int32), ("best_len", C.c_int32), ("exact_len", C.c_int32),
                ("exact_same_as_best", C.c_int32), ("recycled_merges", C.c_uint32), ("max_width_seen", C.c_uint32),
                ("arena_off", C.c_uint64), ("arena_bytes", C.c_uint64), ("nodes_expanded", C.c_uint64),
                ("arcs", C.c_uint64), ("layers", C.c_uint64), ("path_off", C.c_uint64), ("exact_off", C.c_uint64),
                ("cs_state_off", C.c_uint64), ("cs_value_off", C.c_uint64), ("cs_ub_off", C.c_uint64),
                ("cs_path_off", C.c_uint64), ("pool_off", C.c_uint64),
                ("cs_depth_off", C.c_uint64), ("cs_path_stride", C.c_int32), ("cache_hits", C.c_uint32), ("cs_lvar_off", C.c_uint64),
                ("phase_clk", C.c_uint64 * 32)]


def parse_result(res, arena_ptr, ws, depth0):
    """DDResult + arena -> the canonical record of tests/oracle_binding._canon (plus paths)."""
    base = arena_ptr + res.arena_off

    def arr(off, count, ctype, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (ctype * count).from_address(base + off)
        return np.frombuffer(buf, dtype=dtype).copy()

    k = res.n_cutset
    lel = res.cs_path_stride
    cs_depth = arr(res.cs_depth_off, k, C.c_int32, np.int32) if res.cs_depth_off else None
    cs_states = arr(res.cs_state_off, k * ws, C.c_uint64, np.uint64)
    cs_value = arr(res.cs_value_off, k, C.c_int32, np.int32)
    cs_ub = arr(res.cs_ub_off, k, C.c_int32, np.int32)
    if res.cs_lvar_off and k:   # IN_PATH_BITS: bit rows + the variables once -> the u32 rows (variable << 1 | bit), node first
        pw = (lel + 63) // 64
        bits = arr(res.cs_path_off, k * pw, C.c_uint64, np.uint64).reshape(k, pw)
        lvar = arr(res.cs_lvar_off, lel, C.c_uint32, np.uint32)
        cs_paths = np.zeros((k, lel), np.uint32)
        for j in range(lel):
            tr = lel - 1 - j
            cs_paths[:, j] = (lvar[tr] << np.uint32(1)) | ((bits[:, tr >> 6] >> np.uint64(tr & 63)) & np.uint64(1)).astype(np.uint32)
    else:
        cs_paths = arr(res.cs_path_off, k * lel, C.c_uint32, np.uint32).reshape(k, lel) if k else np.zeros((0, lel), np.uint32)
    best_path = arr(res.path_off, res.best_len, C.c_uint32, np.uint32)
    exact_path = arr(res.exact_off, res.exact_len, C.c_uint32, np.uint32)
    cut = sorted((tuple(int(x) for x in cs_states[i * ws:(i + 1) * ws]), int(cs_value[i]), int(cs_ub[i]),
                  depth0 + (int(cs_depth[i]) if cs_depth is not None else lel)) for i in range(k))
    return {
        "status": res.status, "comp_type": res.comp_type,
        "is_exact": bool(res.is_exact) or bool(res.has_exact_best_path),
        "best_value": res.best_value if res.has_best else None,
        "best_exact_value": res.best_exact_value if res.has_best_exact else None,
        "nodes_expanded": res.nodes_expanded, "arcs": res.arcs, "layers": res.layers,
        "cutset": cut,
        "cutset_raw": (cs_states.reshape(k, ws) if k else np.zeros((0, ws), np.uint64), cs_value, cs_ub, cs_paths),
        "best_path": [(int(x) >> 1, int(x) & 1) for x in best_path],
        "exact_path": [(int(x) >> 1, int(x) & 1) for x in exact_path],
        "lel": res.lel, "n_layers": res.n_layers, "recycled_merges": res.recycled_merges, "cache_hits": res.cache_hits,
        "cs_depth": cs_depth,
    }
