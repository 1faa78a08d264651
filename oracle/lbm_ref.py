"""ctypes loader for oracle/build/liblbm_ref.so (TEST INFRASTRUCTURE, never the product path).

The library is the plain-C restatement in oracle/lbm_ref.c of the reference's numpy path
(/root/reference/latticeboltzmann.py:87-123,180-183; order latticeboltzmann.jl:107-120).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liblbm_ref.so")
_lib = None

BOUNDARY = {"reference": 0, "periodic": 1}


def build():
    subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        bp = ctypes.POINTER(ctypes.c_uint8)
        L.lbmref_run.argtypes = [fp, fp, bp, ctypes.c_int, ctypes.c_int, ctypes.c_double, fp,
                                 ctypes.c_int, ctypes.c_int]
        L.lbmref_run.restype = None
        L.lbmref_moments.argtypes = [fp, ctypes.c_long, fp, fp]
        L.lbmref_moments.restype = None
        L.lbmref_equilibrium.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, fp]
        L.lbmref_equilibrium.restype = None
        L.lbmref_collide.argtypes = [fp, bp, ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.lbmref_collide.restype = None
        L.lbmref_stream_reflect.argtypes = [fp, fp, bp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int]
        L.lbmref_stream_reflect.restype = None
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _b(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def run(cells, blocked, steps, omega_free, surroundings, boundary="reference"):
    """Advance float32 C-contiguous `cells` (N,M,9) in place by `steps`; returns cells."""
    assert cells.dtype == np.float32 and cells.flags.c_contiguous
    N, M = cells.shape[:2]
    blk = np.ascontiguousarray(blocked, dtype=np.uint8)
    surr = np.ascontiguousarray(surroundings, dtype=np.float32)
    tmp = np.empty_like(cells)
    lib().lbmref_run(_f(cells), _f(tmp), _b(blk), N, M, float(omega_free), _f(surr),
                     BOUNDARY[boundary], steps)
    return cells


def moments(cells):
    assert cells.dtype == np.float32 and cells.flags.c_contiguous
    N, M = cells.shape[:2]
    rho = np.empty((N, M), np.float32)
    u = np.empty((N, M, 2), np.float32)
    lib().lbmref_moments(_f(cells), N * M, _f(rho), _f(u))
    return u, rho


def equilibrium(u0, u1, p):
    out = np.empty(9, np.float32)
    lib().lbmref_equilibrium(np.float32(u0), np.float32(u1), np.float32(p), _f(out))
    return out


def collide(cells, blocked, omega_free):
    N, M = cells.shape[:2]
    blk = np.ascontiguousarray(blocked, dtype=np.uint8)
    lib().lbmref_collide(_f(cells), _b(blk), N, M, float(omega_free))
    return cells


def stream_reflect(g, blocked, surroundings, boundary="reference"):
    N, M = g.shape[:2]
    blk = np.ascontiguousarray(blocked, dtype=np.uint8)
    surr = np.ascontiguousarray(surroundings, dtype=np.float32)
    f = np.empty_like(g)
    lib().lbmref_stream_reflect(_f(g), _f(f), _b(blk), N, M, _f(surr), BOUNDARY[boundary])
    return f
