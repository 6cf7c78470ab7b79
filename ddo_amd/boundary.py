"""The CALLER side of the drop-in boundary as the reference runs it: T worker threads, one DecisionDiagram each, every thread
looping process_one_node (parallel.rs:391-437, 576-602) through plain `ddo_mdd_compile` / `ddo_mdd_drain_cutset`.  The loop
itself is C++ (tools/b1_driver.cpp -> ddo_amd/_build/libddo_b1.so, a client of libddo_hip.so like any other); this module binds
it for bench.py's `boundary_b1` block and for tests/test_gpu_boundary_b1.py.  Measurement / test support."""
import ctypes as C
import os

import numpy as np

from .binding import DdoError, lib

_B1 = None
MASK64 = (1 << 64) - 1


class Digest(C.Structure):
    _fields_ = [("status", C.c_int32), ("is_exact", C.c_int32), ("has_best", C.c_int32), ("has_best_exact", C.c_int32),
                ("best_value", C.c_int64), ("best_exact_value", C.c_int64), ("nodes_expanded", C.c_uint64), ("arcs", C.c_uint64),
                ("layers", C.c_uint64), ("n_cutset", C.c_uint64), ("cutset_hash", C.c_uint64)]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class _Config(C.Structure):
    _fields_ = [("device", C.c_int), ("cutset_type", C.c_int), ("width", C.c_size_t), ("threads", C.c_int), ("items", C.c_uint64),
                ("best_lb", C.c_int64)]


class Totals(C.Structure):
    _fields_ = [("seconds", C.c_double), ("compiles", C.c_uint64), ("nodes_expanded", C.c_uint64), ("arcs", C.c_uint64),
                ("layers", C.c_uint64), ("cutset_nodes", C.c_uint64), ("path_decisions", C.c_uint64), ("mismatches", C.c_uint64),
                ("errors", C.c_uint64), ("launches", C.c_uint64), ("requests", C.c_uint64), ("kernel_ms", C.c_double), ("compile_s", C.c_double), ("drain_s", C.c_double)]

    def asdict(self):
        return {k: (float if k in ("seconds", "kernel_ms", "compile_s", "drain_s") else int)(getattr(self, k)) for k, _ in self._fields_}


def _b1():
    global _B1
    if _B1 is None:
        lib()   # libddo_hip.so first (the driver links against it)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libddo_b1.so")
        if not os.path.exists(path):
            raise DdoError(f"{path} is missing: run `make -C ddo_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.b1_run.argtypes = [C.c_void_p, C.POINTER(_Config), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Digest),
                             C.POINTER(Digest), C.POINTER(Totals)]
        _B1 = L
    return _B1


def mix64(x):
    x &= MASK64
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & MASK64
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & MASK64
    x ^= x >> 33
    return x


def cutset_hash(cut):
    """the driver's order-independent checksum of a cut-set [(state words.., value, ub, depth)] (tools/b1_driver.cpp: drain_cb)"""
    total = 0
    for state, value, ub, depth in cut:
        h = 0x9e3779b97f4a7c15
        for w in state:
            h = mix64(h ^ int(w))
        for v in (value, ub, depth):
            h = mix64(h ^ (int(v) & MASK64))
        total = (total + h) & MASK64
    return total


def run_b1(model, states, values, depths, width, threads, items, best_lb, device=0, cutset_type=1):
    """T = `threads` worker threads draw `items` work items (item k = sub-problem k % len(values)) and run the reference's
    process_one_node on each.  Returns (totals dict, [restricted digest per sub-problem], [relaxed digest or None])."""
    states = np.ascontiguousarray(states, dtype=np.uint64)
    values = np.ascontiguousarray(values, dtype=np.int64)
    depths = np.ascontiguousarray(depths, dtype=np.int64)
    n = len(values)
    assert states.shape == (n, model.ws)
    cfg = _Config(device, cutset_type, int(width), int(threads), int(items), int(best_lb))
    r0, r1 = (Digest * n)(), (Digest * n)()
    tot = Totals()
    rc = _b1().b1_run(model._h, C.byref(cfg), n, states.ctypes.data_as(C.c_void_p), values.ctypes.data_as(C.c_void_p),
                      depths.ctypes.data_as(C.c_void_p), r0, r1, C.byref(tot))
    if rc != 0:
        raise DdoError(f"b1_run rc={rc}")
    return tot.asdict(), [d.asdict() for d in r0], [None if d.status == -1000 else d.asdict() for d in r1]
