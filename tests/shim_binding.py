"""Builds and binds tests/shim/hip_mdd_shim.cpp: the ORACLE's restatement of the reference's solvers (SequentialSolver / ParallelSolver,
NoDupFringe, MaxUB, widths) with the device engine plugged in as their `DecisionDiagram` through the C ABI -- the compiled twin of
hip_mdd/src/lib.rs.  TEST INFRASTRUCTURE (it includes oracle/): never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class ShimOut(C.Structure):
    _fields_ = [("has_value", C.c_int), ("is_exact", C.c_int), ("best_value", C.c_int64), ("best_lb", C.c_int64), ("best_ub", C.c_int64),
                ("explored", C.c_uint64), ("nodes_expanded", C.c_uint64), ("arcs", C.c_uint64), ("layers", C.c_uint64), ("compiles", C.c_uint64),
                ("wall_s", C.c_double), ("launches", C.c_uint64), ("requests", C.c_uint64), ("n_solution", C.c_int)]


def build_shim():
    src = os.path.join(HERE, "shim", "hip_mdd_shim.cpp")
    out = os.path.join(HERE, "shim", "libddo_shim.so")
    libdir = os.path.join(ROOT, "ddo_amd", "_build")
    deps = [src, os.path.join(ROOT, "oracle", "ddo_oracle.hpp"), os.path.join(ROOT, "oracle", "models.hpp"), os.path.join(ROOT, "include", "ddo_hip.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unused-function", "-o", tmp, src,
                        "-L" + libdir, "-lddo_hip", "-Wl,-rpath," + libdir], check=True)
        os.replace(tmp, out)
    return out


_L = None


def shim_misp_solve(path, width=0, nthreads=0, device=0, timeout_s=0.0, pooled=False, cache_entries=0):
    """the reference's solver (oracle restatement) over HipMdd -- with cache_entries > 0 behind HipCache, the device-side SimpleCache;
    returns a dict like tests.oracle_binding.MispInstance.solve"""
    global _L
    if _L is None:
        _L = C.CDLL(build_shim())
        _L.shim_misp_solve_ex.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64, C.POINTER(ShimOut), C.c_void_p]
    out = ShimOut()
    sol = np.zeros(4096, dtype=np.int64)
    rc = _L.shim_misp_solve_ex(path.encode(), int(width), int(nthreads), int(device), float(timeout_s), 1 if pooled else 0, int(cache_entries),
                               C.byref(out), sol.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("shim_misp_solve failed (message on stderr)")
    d = {k: getattr(out, k) for k, _ in ShimOut._fields_}
    d["solution"] = [(int(sol[2 * i]), int(sol[2 * i + 1])) for i in range(out.n_solution)]
    return d
