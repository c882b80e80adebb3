"""
Drop-in for the reference's ``solve_triangular.py``: ``solve_triangular(L, b)`` returns
``|| L^-1 b ||^2`` (NOT the solution vector), solve_triangular.py:5-14.  ``b`` may also be a
batch ``[nrhs, n]`` -> ``float64[nrhs]`` (one FP64-MFMA blocked solve for all of them).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def solve_triangular(Lmat, b):
    Lm = np.ascontiguousarray(Lmat, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    single = bb.ndim == 1
    bb = np.atleast_2d(bb)
    n = Lm.shape[0]
    if Lm.ndim != 2 or Lm.shape[1] < n or bb.shape[1] != n:
        raise ValueError("L must be (n, n) and b (n,) or (nrhs, n)")
    out = np.empty(bb.shape[0])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.check(L.lib().cf_solve_triangular(p(Lm), n, Lm.shape[1], p(bb), bb.shape[0], p(out)))
    return float(out[0]) if single else out
