"""ctypes wrapper of oracle/fsw_oracle.c  --  TEST INFRASTRUCTURE ONLY (see the header of fsw_oracle.c)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libfsw_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_PATH):
            raise RuntimeError("C oracle not built: run `make -C oracle` (done by __graft_entry__.build())")
        L = ctypes.CDLL(_PATH)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        L.fsw_oracle_embed.restype = ctypes.c_int
        L.fsw_oracle_embed.argtypes = [vp, i64, ctypes.c_int, vp, vp, vp, i64, vp, vp, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, vp, i64, vp, vp, ctypes.c_int]
        L.fsw_oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def embed(X, rowptr, col, w, V, freqs, s0=0, s1=None, tau=1.0, rows=None, nthreads=0, return_mass=False):
    """float64 embedding of the selected rows and slices [s0, s1); X, V, freqs, w are float32 inputs."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    freqs = np.ascontiguousarray(freqs, dtype=np.float32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    rows = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
    s1 = V.shape[0] if s1 is None else s1
    nrows = rowptr.shape[0] - 1
    R = nrows if rows is None else rows.shape[0]
    out = np.empty((R, s1 - s0), dtype=np.float64)
    mass = np.empty(R, dtype=np.float64)
    rc = lib().fsw_oracle_embed(_p(X), X.shape[0], X.shape[1], _p(rowptr), _p(col), _p(w), nrows, _p(V), _p(freqs), s0, s1,
                                float(tau), _p(rows), R, _p(out), _p(mass), nthreads)
    if rc != 0:
        raise MemoryError("fsw_oracle_embed failed")
    return (out, mass) if return_mass else out


def max_threads():
    return lib().fsw_oracle_max_threads()
