"""Identity stand-in for numba's decorators, used ONLY by tests/golden/generate_golden.py.

numba is not installed in the build container.  The reference decorates its likelihood functions
with plain ``@njit`` / ``@njit(parallel=True)`` (no fastmath, no signatures), which are
semantically transparent, so the functions run unchanged as ordinary numpy code.
"""


def njit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


jit = njit
prange = range
