"""Builds and binds the lock-step host emulation of the device kernel (tests/emul).
TEST INFRASTRUCTURE: lets the CPU-only suite exercise the workgroup logic of
ddo_amd/csrc/misp_dd_core.hpp; never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from tests.dd_wire import DDInput, DDResult, parse_result, MAX_WS

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build_emul():
    src = os.path.join(HERE, "emul", "emul_capi.cpp")
    out = os.path.join(HERE, "emul", "libddo_emul.so")
    deps = [src] + [os.path.join(ROOT, "ddo_amd", "csrc", f) for f in ("misp_dd_core.hpp", "misp_dd_inplace.hpp", "dd_types.h", "engine.hpp",
                                                                          "dd_thresholds.hpp", "dd_tsptw.hpp")]
    def stale():
        return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)

    if stale():
        import fcntl
        with open(out + ".lock", "w") as lk:   # pytest-xdist workers: one of them builds, the others wait and find it fresh
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                tmp = out + ".%d.tmp" % os.getpid()
                subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-Wall", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", tmp, src],
                               check=True)
                os.replace(tmp, out)
    return out


class Emul:
    def __init__(self, n, rows, weights, max_width, nthreads=256, arena_bytes=64 << 20, engine=1):
        L = C.CDLL(build_emul())
        L.emul_create.restype = C.c_void_p
        L.emul_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.emul_destroy.argtypes = [C.c_void_p]
        L.emul_state_words.argtypes = [C.c_void_p]
        L.emul_compile.argtypes = [C.c_void_p, C.POINTER(DDInput), C.POINTER(DDResult), C.POINTER(C.c_void_p)]
        L.emul_sizeof_input.restype = C.c_uint64
        L.emul_sizeof_result.restype = C.c_uint64
        assert L.emul_sizeof_input() == C.sizeof(DDInput), (L.emul_sizeof_input(), C.sizeof(DDInput))
        assert L.emul_sizeof_result() == C.sizeof(DDResult), (L.emul_sizeof_result(), C.sizeof(DDResult))
        self.L = L
        self.n = n
        self.ws_in = (n + 63) // 64
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        weights = np.ascontiguousarray(weights, dtype=np.int64)
        self.h = L.emul_create(n, rows.ctypes.data_as(C.c_void_p), weights.ctypes.data_as(C.c_void_p), max_width, nthreads,
                               arena_bytes, engine)
        if not self.h:
            raise RuntimeError("emul_create failed")
        self.ws = L.emul_state_words(self.h)

    def __del__(self):
        try:
            self.L.emul_destroy(self.h)
        except Exception:
            pass

    def keep_layers(self, on=True, cache_entries=0):
        """every layer kept (frontier cut-set, thresholds); cache_entries > 0: a fresh, empty SimpleCache table"""
        self.L.emul_set_keep_layers.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        self.L.emul_set_keep_layers(self.h, 1 if on else 0, int(cache_entries))

    def pooled(self, on=True):
        """engine 2: compile Pooled decision diagrams (mdd/pooled.rs) -- run_dd2<WS, DEEP, POOLED = 1>"""
        self.L.emul_set_pooled.argtypes = [C.c_void_p, C.c_int]
        self.L.emul_set_pooled(self.h, 1 if on else 0)

    def pooled_cache(self, entries):
        """Pooled behind a fresh, empty SimpleCache (pooled.rs:467-535, 662-680); 0: EmptyCache"""
        self.L.emul_set_pooled_cache.argtypes = [C.c_void_p, C.c_uint64]
        self.L.emul_set_pooled_cache(self.h, int(entries))

    def dominance(self, cap):
        """a fresh SimpleDominanceChecker with `cap` pairs per depth (0: none); needs keep_layers(True)"""
        self.L.emul_set_dominance.argtypes = [C.c_void_p, C.c_uint32]
        self.L.emul_set_dominance(self.h, int(cap))

    def cache_used(self):
        self.L.emul_cache_used.restype = C.c_uint64
        self.L.emul_cache_used.argtypes = [C.c_void_p]
        return self.L.emul_cache_used(self.h)

    def compile(self, comp_type, width, best_lb, state, value, depth, flags=4):
        inp = DDInput()
        inp.comp_type = comp_type
        inp.flags = flags
        inp.width = int(width)
        inp.value = int(value)
        inp.depth = int(depth)
        inp.best_lb = int(max(min(best_lb, (1 << 62)), -(1 << 62)))
        inp.src_off = (1 << 64) - 1
        for k in range(MAX_WS):
            inp.state[k] = int(state[k]) if k < len(state) else 0
        res = (DDResult * 2)()
        arena = C.c_void_p()
        rc = self.L.emul_compile(self.h, C.byref(inp), res, C.byref(arena))
        if rc != 0:
            raise RuntimeError(f"emul_compile rc={rc}")
        out = []
        for r in res:
            if r.status == 77:
                out.append(None)
                continue
            d = parse_result(r, arena.value, self.ws, depth)
            # trim padded state words
            d["cutset"] = sorted((s[:self.ws_in], v, u, dp) for (s, v, u, dp) in d["cutset"])
            out.append(d)
        return out


class ModelEmul(Emul):
    """Engine-1 emulation bound to a model descriptor of the product library (any kind: MISP, knapsack, MCP, MAX2SAT)."""

    def __init__(self, model, max_width, nthreads=256, arena_bytes=16 << 20):   # engine 1 needs >= 256 threads (256-bin scans)
        L = C.CDLL(build_emul())
        L.emul_create_model.restype = C.c_void_p
        L.emul_create_model.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64]
        L.emul_destroy.argtypes = [C.c_void_p]
        L.emul_state_words.argtypes = [C.c_void_p]
        L.emul_compile.argtypes = [C.c_void_p, C.POINTER(DDInput), C.POINTER(DDResult), C.POINTER(C.c_void_p)]
        self.L = L
        self.model = model            # keeps the descriptor (and the tables the emulator points into) alive
        self.n = model.n
        self.ws_in = model.ws
        self.h = L.emul_create_model(model._h, max_width, nthreads, arena_bytes)
        if not self.h:
            raise RuntimeError("emul_create_model failed")
        self.ws = L.emul_state_words(self.h)

    def compile_root(self, comp_type, width, best_lb=-(1 << 40)):
        return self.compile(comp_type, width, best_lb, self.model.initial_state(), self.model.initial_value(), 0)
