"""
Drop-in for the reference's ``interpolator.py`` (same function names and argument order), evaluated
by the HIP ``interp_kernel`` through the C-ABI.  interpolator.py:111-119.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _prep(*arrs):
    return [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def interp_hermite(xq, x, y, y_prime):
    """Cubic Hermite with user slopes; linear extrapolation outside the grid (exact=True)."""
    xq, x, y, yp = _prep(np.atleast_1d(xq), x, y, y_prime)
    if not (x.size == y.size == yp.size):
        raise ValueError("x, y, y_prime must have the same length")
    out = np.empty_like(xq)
    L.check(L.lib().cf_interp_hermite(_p(xq), xq.size, _p(x), _p(y), _p(yp), x.size, _p(out)))
    return out


def interp_pchip(xq, x, y):
    """PCHIP (Fritsch-Carlson slopes), clamped to the end values outside the grid."""
    xq, x, y = _prep(np.atleast_1d(xq), x, y)
    if x.size != y.size:
        raise ValueError("x and y must have the same length")
    out = np.empty_like(xq)
    L.check(L.lib().cf_interp_pchip(_p(xq), xq.size, _p(x), _p(y), x.size, _p(out)))
    return out
