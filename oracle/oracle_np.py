"""
oracle_np — CPU restatement (numpy) of the reference's walker log-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker.  The product path (``cosmology-model-fit_amd``) never imports this module and fails
loudly when its HIP library is missing.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against golden
vectors produced by running the reference's own functions in this container
(``tests/golden/generate_golden.py``; the reference's ``@njit`` functions executed as plain
numpy through an identity ``njit`` decorator because numba is not installed).  The large SN
covariances are missing from the reference snapshot (``.MISSING_LARGE_BLOBS``), so SN-block
fixtures use the real redshift/magnitude columns with a seeded synthetic SPD covariance.

Each function cites the reference lines it restates (paths relative to the reference repo).
The operation ORDER follows the reference (left-to-right numpy expressions, sequential cumsum,
forward substitution) so that results agree to a few ulp, not merely to the 1e-10 parity bar.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

C_KM_S = 299792.458  # scipy.constants.c / 1000   (sn/pantheon.py:12)

# f_DE families ------------------------------------------------------------------------------
FDE_LCDM, FDE_WCDM, FDE_THAWING, FDE_CPL = 0, 1, 2, 3
EZ_LATE_FLAT, EZ_PHYSICAL = 0, 1


# ---------------------------------------------------------------------------------------------
# interpolator.py
# ---------------------------------------------------------------------------------------------
def pchip_slopes(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Fritsch-Carlson slopes, interpolator.py:5-68 (same branch logic, same formulas)."""
    n = len(x)
    if n < 2:
        return np.zeros(n)
    h = x[1:] - x[:-1]
    delta = (y[1:] - y[:-1]) / h
    d = np.zeros(n)
    if n == 2:
        d[:] = delta[0]
        return d
    dl, dr = delta[:-1], delta[1:]
    hl, hr = h[:-1], h[1:]
    ok = (dl != 0.0) & (dr != 0.0) & (dl * dr > 0.0)
    w1 = 2.0 * hr + hl
    w2 = hr + 2.0 * hl
    with np.errstate(divide="ignore", invalid="ignore"):
        inner = (w1 + w2) / (w1 / dl + w2 / dr)
    d[1:-1] = np.where(ok, inner, 0.0)

    # start point, interpolator.py:41-50
    d0 = ((2 * h[0] + h[1]) * delta[0] - h[0] * delta[1]) / (h[0] + h[1])
    if delta[0] == 0.0 or np.sign(d0) != np.sign(delta[0]):
        d[0] = 0.0
    elif (np.sign(delta[0]) != np.sign(delta[1])) and (abs(d0) > abs(3 * delta[0])):
        d[0] = 3 * delta[0]
    else:
        d[0] = d0
    # end point, interpolator.py:52-66
    dn = ((2 * h[n - 2] + h[n - 3]) * delta[n - 2] - h[n - 2] * delta[n - 3]) / (h[n - 2] + h[n - 3])
    if delta[n - 2] == 0.0 or np.sign(dn) != np.sign(delta[n - 2]):
        d[n - 1] = 0.0
    elif (np.sign(delta[n - 2]) != np.sign(delta[n - 3])) and (abs(dn) > abs(3 * delta[n - 2])):
        d[n - 1] = 3 * delta[n - 2]
    else:
        d[n - 1] = dn
    return d


def _cubic_interp(xq, x, y, d, exact: bool):
    """interpolator.py:71-108.  searchsorted(..., 'left') - 1 interval rule (line 94); outside the
    grid: linear extrapolation when ``exact`` (lines 87-92) else clamp (lines 80-85)."""
    xq = np.atleast_1d(np.asarray(xq, dtype=np.float64))
    out = np.empty_like(xq)
    lo = xq <= x[0]
    hi = xq >= x[-1]
    if exact:
        out[lo] = y[0] + d[0] * (xq[lo] - x[0])
        out[hi] = y[-1] + d[-1] * (xq[hi] - x[-1])
    else:
        out[lo] = y[0]
        out[hi] = y[-1]
    mid = ~(lo | hi)
    xi = xq[mid]
    i = np.searchsorted(x, xi) - 1
    h_i = x[i + 1] - x[i]
    t = (xi - x[i]) / h_i
    t2 = t * t
    t3 = t2 * t
    h00 = 2 * t3 - 3 * t2 + 1
    h10 = t3 - 2 * t2 + t
    h01 = -2 * t3 + 3 * t2
    h11 = t3 - t2
    out[mid] = h00 * y[i] + h10 * h_i * d[i] + h01 * y[i + 1] + h11 * h_i * d[i + 1]
    return out


def interp_hermite(xq, x, y, y_prime):
    """interpolator.py:117-119."""
    return _cubic_interp(xq, x, y, y_prime, True)


def interp_pchip(xq, x, y):
    """interpolator.py:111-114."""
    return _cubic_interp(xq, x, y, pchip_slopes(x, y), False)


# ---------------------------------------------------------------------------------------------
# solve_triangular.py
# ---------------------------------------------------------------------------------------------
def solve_triangular_chi2(L: np.ndarray, b: np.ndarray) -> float:
    """solve_triangular.py:5-14: forward substitution, returns y.y (NOT y). Only L[i,:i+1] is read."""
    n = len(b)
    y = np.empty(n)
    for i in range(n):
        y[i] = (b[i] - np.dot(L[i, :i], y[:i])) / L[i, i]
    return float(np.dot(y, y))


# ---------------------------------------------------------------------------------------------
# likelihood descriptor (oracle side; independent of the product's cf_desc)
# ---------------------------------------------------------------------------------------------
@dataclass
class Slot:
    idx: int = -1
    scale: float = 1.0
    fixed: float = 0.0

    def get(self, theta):
        return self.scale * theta[self.idx] if self.idx >= 0 else self.fixed


@dataclass
class Likelihood:
    ndim: int
    z_max: float
    n_grid: int = 4000
    ez_model: int = EZ_LATE_FLAT
    fde: int = FDE_LCDM
    c: float = C_KM_S
    # parameter slots
    offset: Slot = field(default_factory=Slot)
    H0: Slot = field(default_factory=Slot)
    Om: Slot = field(default_factory=Slot)
    obh2: Slot = field(default_factory=Slot)
    och2: Slot = field(default_factory=Slot)
    w0: Slot = field(default_factory=lambda: Slot(fixed=-1.0))
    wa: Slot = field(default_factory=Slot)
    v: Slot = field(default_factory=Slot)
    rd: Slot = field(default_factory=Slot)
    # SN block
    z_cmb: Optional[np.ndarray] = None
    z_hel: Optional[np.ndarray] = None
    obs: Optional[np.ndarray] = None
    step: Optional[np.ndarray] = None  # per-SN sign/weight; None -> from z_turn
    z_turn: float = 0.15
    chol: Optional[np.ndarray] = None
    # priors
    bounds: Optional[np.ndarray] = None
    gauss: Sequence = ()  # (idx, mean, sigma) on the log-prior

    def __post_init__(self):
        # z_grid / dz exactly as sn/pantheon.py:16-17
        self.z_grid = np.linspace(0, self.z_max, num=self.n_grid)
        self.dz = np.diff(self.z_grid)
        if self.z_cmb is not None and self.step is None:
            self.step = np.where(self.z_cmb <= self.z_turn, 1.0, -1.0)  # sn/pantheon.py:46


def f_de(lk: Likelihood, z, theta):
    """Dark-energy density ratio; one formula per reference variant."""
    zp1 = 1.0 + z
    if lk.fde == FDE_LCDM:
        return 1.0
    w0 = lk.w0.get(theta)
    if lk.fde == FDE_WCDM:  # sn/pantheon_and_sh0es.py:26-28
        return zp1 ** (3 * (1 + w0))
    if lk.fde == FDE_THAWING:  # bao/desi.py:26-28, sn/pantheon.py:22-25
        cubed = zp1 * zp1 * zp1
        return (2 * cubed / ((1.0 + w0) + (1.0 - w0) * cubed)) ** 2
    if lk.fde == FDE_CPL:  # bao/desi_fs_lya_cmb.py:19-22
        wa = lk.wa.get(theta)
        return zp1 ** (3 * (1 + w0 + wa)) * np.exp(-3 * wa * z / zp1)
    raise ValueError(lk.fde)


def H_z(lk: Likelihood, z, theta):
    """sn/pantheon.py:28-31 (late-time flat).  (1+z)**3 is evaluated as multiplies, as numba does."""
    H0 = lk.H0.get(theta)
    if lk.ez_model == EZ_LATE_FLAT:
        Om = lk.Om.get(theta)
        zp1 = 1.0 + z
        cubed = zp1 * zp1 * zp1
        if lk.fde == FDE_LCDM:
            return H0 * np.sqrt(Om * cubed + (1.0 - Om))
        return H0 * np.sqrt(Om * cubed + (1.0 - Om) * f_de(lk, z, theta))
    raise NotImplementedError("EZ_PHYSICAL arrives with the CMB block")


def dm_grid(lk: Likelihood, theta):
    """Cumulative trapezoid, sn/pantheon.py:35-39 / bao/desi_cmb_des5y.py:60-66."""
    dh_grid = lk.c / H_z(lk, lk.z_grid, theta)
    dh = (dh_grid[:-1] + dh_grid[1:]) / 2
    cum_dm = np.zeros(lk.z_grid.size)
    cum_dm[1:] = np.cumsum(dh * lk.dz)
    return cum_dm, dh_grid


def sn_parts(lk: Likelihood, theta):
    """DM(z_cmb), mu_corr, mu_theory, residual: sn/pantheon.py:43-61."""
    cum_dm, dh_grid = dm_grid(lk, theta)
    DM = interp_hermite(lk.z_cmb, lk.z_grid, cum_dm, dh_grid)
    v_km_s = 100 * lk.v.get(theta) * lk.step
    z_pec = v_km_s / lk.c
    z_cosmo = -1.0 + (1.0 + lk.z_cmb) / (1.0 + z_pec)
    mu_corr = 5.0 * np.log10(interp_hermite(z_cosmo, lk.z_grid, cum_dm, dh_grid) / DM)
    mu_theory = 25.0 + 5 * np.log10((1.0 + lk.z_hel) * DM)
    delta = lk.obs - lk.offset.get(theta) - mu_corr - mu_theory
    return DM, mu_corr, mu_theory, delta


def chi_squared(lk: Likelihood, theta) -> float:
    """sn/pantheon.py:57-61."""
    theta = np.asarray(theta, dtype=np.float64)
    *_, delta = sn_parts(lk, theta)
    return solve_triangular_chi2(lk.chol, delta)


def log_prior(lk: Likelihood, theta) -> float:
    """sn/pantheon.py:80-85: strict box, flat normalisation, optional Gaussian terms."""
    theta = np.asarray(theta, dtype=np.float64)
    b = lk.bounds
    if b is not None:
        if not np.all((b[:, 0] < theta) & (theta < b[:, 1])):
            return -np.inf
        lp = -np.sum(np.log(b[:, 1] - b[:, 0]))
    else:
        lp = 0.0
    for idx, mean, sigma in lk.gauss:
        lp = lp - 0.5 * (theta[idx] - mean) ** 2 / sigma**2
    return float(lp)


def log_probability(lk: Likelihood, theta) -> float:
    """sn/pantheon.py:88-97: the likelihood is NOT evaluated outside the box."""
    lp = log_prior(lk, theta)
    if np.isinf(lp):
        return -np.inf
    return lp - 0.5 * chi_squared(lk, theta)


def log_probs_vectorized(lk: Likelihood, batch) -> np.ndarray:
    """bao/desi.py:100-106 (row loop); float64 out (superset of the reference's float32)."""
    return np.array([log_probability(lk, row) for row in np.atleast_2d(batch)])
