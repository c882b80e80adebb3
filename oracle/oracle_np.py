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
    fcc: Slot = field(default_factory=lambda: Slot(fixed=1.0))
    s8: Slot = field(default_factory=Slot)      # sigma_8(z = 0) of the growth-rate block
    fs8err: Slot = field(default_factory=lambda: Slot(fixed=1.0))  # its error-rescale factor f_err
    lin: Slot = field(default_factory=Slot)  # amplitude of the per-SN linear magnitude term (bulk-flow correction)
    v2: Slot = field(default_factory=Slot)   # second / third velocity component of a direction-dependent peculiar velocity
    v3: Slot = field(default_factory=Slot)
    om_mode: int = 0  # 1: slot Om holds omega_m = Omega_m h^2 (bao/desi_omh2.py:18-20)
    # SN block
    z_cmb: Optional[np.ndarray] = None
    z_hel: Optional[np.ndarray] = None
    obs: Optional[np.ndarray] = None
    step: Optional[np.ndarray] = None  # per-SN sign/weight; None -> from z_turn
    z_turn: float = 0.15
    chol: Optional[np.ndarray] = None
    has_vstep: bool = True  # False: no peculiar-velocity step at all (z_cosmo = z_cmb, mu_corr = 0)
    fixed_mu: Optional[np.ndarray] = None  # per SN: NaN -> mu_theory, else this distance modulus (SH0ES calibrators)
    lin_coef: Optional[np.ndarray] = None  # per SN: offset_i = offset + lin * lin_coef[i] (bao/desi_cmb_pantheon_H0trgb.py:102-106)
    dirs: Optional[np.ndarray] = None  # [N, 3] unit vectors: v_los = n . (v, v2, v3), weight = step (sn/pantheon_dipole_xyz.py:50-60)
    # cosmic-chronometer block: H(z) data with explicit inverse covariance and error-rescale parameter f_cc
    cc_z: Optional[np.ndarray] = None
    cc_h: Optional[np.ndarray] = None
    cc_inv_cov: Optional[np.ndarray] = None
    cc_logdet: float = 0.0
    # growth-rate block (fs8/fs8.py): data, covariance (the reference solves with its Cholesky factor or multiplies by the
    # explicit inverse), fiducial H(z) D_M(z) of the Alcock-Paczynski factor, the a grid handed to solve_ivp as t_eval
    fs8_z: Optional[np.ndarray] = None
    fs8_val: Optional[np.ndarray] = None
    fs8_inv_cov: Optional[np.ndarray] = None
    fs8_fid: Optional[np.ndarray] = None
    fs8_a_span: Optional[np.ndarray] = None
    # radiation + massive-neutrino constants of a cmb.data_*_compression module (EZ_PHYSICAL)
    or_h2: float = 0.0
    omnu_h2: float = 0.0
    o_gamma_h2: float = 0.0
    nu_m0: float = 0.0
    nu_rho0: float = 1.0
    nu_qs_sq: Optional[np.ndarray] = None
    nu_ws: Optional[np.ndarray] = None
    # BAO block
    bao_z: Optional[np.ndarray] = None
    bao_val: Optional[np.ndarray] = None
    bao_qty: Optional[np.ndarray] = None  # 0 DV/rd, 1 DM/rd, 2 DH/rd, 3 F_AP
    bao_inv_cov: Optional[np.ndarray] = None
    bao_dh_exact: bool = False  # False: PCHIP on the dh grid; True: c/H(z) at the datum
    rd_fit: Optional[Sequence] = None  # (b, m, a1..a9) -> r_drag fitting formula; None -> slot `rd`
    rd_wm_late: bool = False  # the fit takes wm = Omega_m h^2 of the late-time flat model (bao/desi_bbn.py:46-60)
    # compressed CMB block
    cmb_mode: int = 0  # 0 none, 1 (R, lA, wb), 2 lA only, 3 (theta*, wb, wm)
    cmb_prior: Optional[np.ndarray] = None
    cmb_inv_cov: Optional[np.ndarray] = None  # 3x3 (mode 2: only [1,1] = 1/var is used)
    zstar_fit: Optional[Sequence] = None  # (s1, s2, b, m)
    gl_x: Optional[np.ndarray] = None
    gl_w: Optional[np.ndarray] = None
    # priors
    bounds: Optional[np.ndarray] = None
    gauss: Sequence = ()  # (idx, mean, sigma) on the log-prior
    chi2_gauss: Sequence = ()  # (idx, mean, sigma) added to chi^2
    cpl_wall: bool = False  # w0 + wa >= 0 -> log L = -1e8   (bao/desi_fs_lya_cmb.py:118-121)
    sn_vel_mult: bool = False  # z_cosmo = max((1 + z_cmb)(1 + z_pec) - 1, 1e-8), bao/desi_pantheon_cc.py:84-87
    cc_f_inverse: bool = False  # chi2_cc * f^-2 and + 2 N ln f in the normalisation, ohd/cc_pantheon.py:64,92
    prior_normalised: bool = True  # False: log prior = 0.0 inside the box, ohd/cc_cmb.py:70-73
    logl_const: float = 0.0  # constant added to log L (fs8/fs8_cmb.py:20,181-183: -0.5 (N ln 2 pi + logdet))

    def __post_init__(self):
        # z_grid / dz exactly as sn/pantheon.py:16-17
        self.z_grid = np.linspace(0, self.z_max, num=self.n_grid)
        self.dz = np.diff(self.z_grid)
        if self.cmb_mode and self.gl_x is None:
            self.gl_x, self.gl_w = np.polynomial.legendre.leggauss(100)  # cmb/...:150
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


def Omnu_z(lk: Likelihood, z):
    """5-node massive-neutrino density, cmb/data_planck_act_compression.py:53-66."""
    zp1 = 1.0 + z
    mz_sq = (lk.nu_m0 / zp1) ** 2
    q, w = lk.nu_qs_sq, lk.nu_ws
    f = [np.sqrt(q[i] + mz_sq) for i in range(5)]
    weighted_sum = f[0] * w[0] + f[1] * w[1] + f[2] * w[2] + f[3] * w[3] + f[4] * w[4]
    zp1_2 = zp1 * zp1
    return zp1_2 * zp1_2 * weighted_sum / lk.nu_rho0


def H_z(lk: Likelihood, z, theta):
    """Late-time flat: sn/pantheon.py:28-31.  Physical densities: bao/desi_cmb_des5y.py:34-54.
    Integer powers are evaluated as multiplies, as numba does."""
    H0 = lk.H0.get(theta)
    zp1 = 1.0 + z
    cubed = zp1 * zp1 * zp1
    if lk.ez_model == EZ_LATE_FLAT:
        Om = lk.Om.get(theta)
        if lk.om_mode:  # bao/desi_omh2.py:18-20: h, Omh2 = params[1] / 100, params[2]; Om = Omh2 / h**2
            h = H0 / 100
            Om = Om / (h * h)
        if lk.fde == FDE_LCDM:
            return H0 * np.sqrt(Om * cubed + (1.0 - Om))
        return H0 * np.sqrt(Om * cubed + (1.0 - Om) * f_de(lk, z, theta))
    h = H0 / 100
    Onu = lk.omnu_h2 / (h * h)
    Or = lk.or_h2 / (h * h)
    Obc = (lk.obh2.get(theta) + lk.och2.get(theta)) / (h * h)
    Ode = 1.0 - Obc - Or - Onu
    radiation_term = Or * (cubed * zp1)
    matter_term = Obc * cubed
    neutrino_term = Onu * Omnu_z(lk, z)
    dark_energy_term = Ode * f_de(lk, z, theta) if lk.fde != FDE_LCDM else Ode
    return H0 * np.sqrt(radiation_term + matter_term + dark_energy_term + neutrino_term)


# ---- fitting formulae, cmb/data_planck_act_compression.py:86-124 (arXiv:2106.00428) -------------------------
def z_star(fit, wb, wm):
    s1, s2, b, m = fit
    wb = wb**b
    wm = wm**m
    return (
        wm**-0.7316314841257655
        + s1 * 391.6723594873167 * wb**0.9368102670600895 * wm**-0.35300106475765136
        + s2 * 937.4224935298015 * wm**0.0192950634264157 * wb**-0.04285000485853785
    )


def r_drag(fit, wb, wm):
    b, m, a1, a2, a3, a4, a5, a6, a7, a8, a9 = fit
    wb = wb**b
    wm = wm**m
    term_A_denominator = (a1 * (wb**a2)) + (a3 * (wb**a4) * (wm**a5)) + (a6 * (wm**a7))
    return 1.0 / term_A_denominator - a8 / (wm**a9)


def cmb_distances(lk: Likelihood, theta):
    """cmb/data_planck_act_compression.py:160-212 (100-node Gauss-Legendre in z and in a)."""
    Ob_h2, Oc_h2 = lk.obh2.get(theta), lk.och2.get(theta)
    Om_h2 = Oc_h2 + Ob_h2 + lk.omnu_h2
    zstar = z_star(lk.zstar_fit, Ob_h2, Om_h2)
    # rs_z
    a_lim = 1.0 / (1.0 + zstar)
    half = a_lim / 2.0
    integral = 0.0
    for i in range(len(lk.gl_x)):
        a = half * lk.gl_x[i] + half
        z = (1.0 / a) - 1.0
        Rb = (3.0 / 4.0) * (Ob_h2 / lk.o_gamma_h2) * a
        integral += lk.gl_w[i] * (lk.c / (a**2 * H_z(lk, z, theta) * np.sqrt(3.0 * (1.0 + Rb))))
    rs_star = half * integral
    # DM_z
    half = zstar / 2.0
    integral = 0.0
    for i in range(len(lk.gl_x)):
        integral += lk.gl_w[i] * (lk.c / H_z(lk, half * lk.gl_x[i] + half, theta))
    DM_star = half * integral
    if lk.cmb_mode == 3:  # cmb/data_early_lcdm_compression.py:206-207
        return np.array([rs_star / DM_star, Ob_h2, Om_h2])
    R = 100 * np.sqrt(Om_h2) * DM_star / lk.c
    lA = np.pi * DM_star / rs_star
    return np.array([R, lA, Ob_h2])


def chi2_cmb(lk: Likelihood, theta) -> float:
    delta = lk.cmb_prior - cmb_distances(lk, theta)
    if lk.cmb_mode == 2:  # bao/desi_des5y_bbn_theta_star.py:110-111
        return float(delta[1] ** 2 * lk.cmb_inv_cov[1, 1])
    return float(delta @ lk.cmb_inv_cov @ delta)  # bao/desi_cmb_des5y.py:126-129


def bao_theory(lk: Likelihood, theta, tables=None):
    """bao/desi_cmb_des5y.py:82-100, bao/desi.py:38-56, bao/desi_cmb.py:79-91."""
    cum_dm, dh_grid = tables if tables is not None else dm_grid(lk, theta)
    z, qty = lk.bao_z, lk.bao_qty
    if lk.rd_fit is not None and lk.rd_wm_late:  # bao/desi_bbn.py:47,60: h, Om, Obh2 -> r_drag(Obh2, Om * h**2)
        h = lk.H0.get(theta) / 100
        rd = r_drag(lk.rd_fit, lk.obh2.get(theta), lk.Om.get(theta) * h**2)
    elif lk.rd_fit is not None:
        Obh2, Och2 = lk.obh2.get(theta), lk.och2.get(theta)
        rd = r_drag(lk.rd_fit, Obh2, Obh2 + Och2 + lk.omnu_h2)
    else:
        rd = lk.rd.get(theta)
    DM = interp_hermite(z, lk.z_grid, cum_dm, dh_grid)
    DH = lk.c / H_z(lk, z, theta) if lk.bao_dh_exact else interp_pchip(z, lk.z_grid, dh_grid)
    out = np.empty(z.size)
    for k in range(z.size):
        if qty[k] == 2:
            out[k] = DH[k] / rd
        elif qty[k] == 1:
            out[k] = DM[k] / rd
        elif qty[k] == 0:
            out[k] = (z[k] * DH[k] * DM[k] ** 2) ** (1 / 3) / rd
        else:
            out[k] = DM[k] / DH[k]
    return out


def chi2_bao(lk: Likelihood, theta, tables=None) -> float:
    delta = lk.bao_val - bao_theory(lk, theta, tables)
    return float(delta @ lk.bao_inv_cov @ delta)


def dm_grid(lk: Likelihood, theta):
    """Cumulative trapezoid, sn/pantheon.py:35-39 / bao/desi_cmb_des5y.py:60-66."""
    dh_grid = lk.c / H_z(lk, lk.z_grid, theta)
    dh = (dh_grid[:-1] + dh_grid[1:]) / 2
    cum_dm = np.zeros(lk.z_grid.size)
    cum_dm[1:] = np.cumsum(dh * lk.dz)
    return cum_dm, dh_grid


def sn_parts(lk: Likelihood, theta, tables=None):
    """DM(z_cmb), mu_corr, mu_theory, residual: sn/pantheon.py:43-61."""
    cum_dm, dh_grid = tables if tables is not None else dm_grid(lk, theta)
    DM = interp_hermite(lk.z_cmb, lk.z_grid, cum_dm, dh_grid)
    if lk.has_vstep:
        if lk.dirs is not None:  # sn/pantheon_dipole_xyz.py:54-57; step = attenuation * survey_mask
            v_los = lk.dirs[:, 0] * lk.v.get(theta) + lk.dirs[:, 1] * lk.v2.get(theta) + lk.dirs[:, 2] * lk.v3.get(theta)
            v_km_s = 100 * v_los * lk.step
        else:
            v_km_s = 100 * lk.v.get(theta) * lk.step
        z_pec = v_km_s / lk.c
        z_cosmo = -1.0 + (1.0 + lk.z_cmb) / (1.0 + z_pec)
        if lk.sn_vel_mult:
            z_cosmo = np.maximum((1.0 + lk.z_cmb) * (1.0 + z_pec) - 1.0, 1e-8)
        mu_corr = 5.0 * np.log10(interp_hermite(z_cosmo, lk.z_grid, cum_dm, dh_grid) / DM)
    else:  # bao/desi_des5y_bbn_theta_star.py:94-97: no step term at all
        mu_corr = np.zeros_like(DM)
    mu_theory = 25.0 + 5 * np.log10((1.0 + lk.z_hel) * DM)
    if lk.fixed_mu is not None:  # sn/pantheon_and_sh0es.py:65
        mu_theory = np.where(np.isnan(lk.fixed_mu), mu_theory, lk.fixed_mu)
    offset = lk.offset.get(theta)
    if lk.lin_coef is not None:  # M = params[0] + v_flow_corr, bao/desi_cmb_pantheon_H0trgb.py:104-105
        offset = offset + lk.lin.get(theta) * lk.lin_coef
    delta = lk.obs - offset - mu_corr - mu_theory
    return DM, mu_corr, mu_theory, delta


def chi2_cc(lk: Likelihood, theta) -> float:
    """bao/desi_union3_cc_theta_star.py:129-130."""
    delta = lk.cc_h - H_z(lk, lk.cc_z, theta)
    f = lk.fcc.get(theta)
    return float(delta @ lk.cc_inv_cov @ delta * (f**-2 if lk.cc_f_inverse else f**2))


# ---- growth rate f sigma_8: fs8/fs8.py:26-120, bao/desi_cmb_union3_fs8.py:27-207 ----------------------------------------
def w_nu_z(lk: Likelihood, z):
    """Massive-neutrino equation of state, cmb/data_planck_act_compression.py:70-83."""
    mz_sq = (lk.nu_m0 / (1.0 + z)) ** 2
    f = [np.sqrt(lk.nu_qs_sq[i] + mz_sq) for i in range(5)]
    w = lk.nu_ws
    numerator = w[0] / f[0] + w[1] / f[1] + w[2] / f[2] + w[3] / f[3] + w[4] / f[4]
    denominator = w[0] * f[0] + w[1] * f[1] + w[2] * f[2] + w[3] * f[3] + w[4] * f[4]
    return (1 / 3) - (1 / 3) * mz_sq * numerator / denominator


def d_fde_dz(lk: Likelihood, z, theta):
    """d f_DE / dz = f_DE * 3 (1 + w(z)) / (1 + z), fs8/fs8.py:26-41 for the thawing form."""
    if lk.fde == FDE_LCDM:
        return 0.0
    w0 = lk.w0.get(theta)
    if lk.fde == FDE_WCDM:
        w = w0
    elif lk.fde == FDE_THAWING:
        w = -1.0 + 2 * (1.0 + w0) / (1.0 + w0 + (1.0 - w0) * (1.0 + z) ** 3)
    else:
        w = w0 + lk.wa.get(theta) * z / (1.0 + z)
    return f_de(lk, z, theta) * 3 * (1.0 + w) / (1.0 + z)


def growth_ODE(a, y, lk: Likelihood, theta):
    """fs8/fs8.py:64-76 (E form) and bao/desi_cmb_union3_fs8.py:127-165 (H form): the same equation."""
    z = 1 / a - 1.0
    H0 = lk.H0.get(theta)
    H_val = H_z(lk, z, theta)
    if lk.ez_model == EZ_LATE_FLAT:
        Om = lk.Om.get(theta)
        if lk.om_mode:
            Om = Om / (H0 / 100) ** 2
        numerator = 3 * Om * (1.0 + z) ** 2 + (1.0 - Om) * d_fde_dz(lk, z, theta)
        om_growth = Om
    else:
        h = H0 / 100
        Obc = (lk.obh2.get(theta) + lk.och2.get(theta)) / h**2
        Or, Onu = lk.or_h2 / h**2, lk.omnu_h2 / h**2
        Ode = 1.0 - Obc - Or - Onu
        d_Omnu_dz = Omnu_z(lk, z) * 3 * (1.0 + w_nu_z(lk, z)) / (1.0 + z)
        numerator = 3 * Obc * (1.0 + z) ** 2 + 4 * Or * (1.0 + z) ** 3 + Onu * d_Omnu_dz + Ode * d_fde_dz(lk, z, theta)
        om_growth = Obc
    dH_da = -numerator * H0**2 / (2 * H_val / (1.0 + z) ** 2)
    delta, d_delta_da = y
    source = 1.5 * om_growth * (H0 / H_val) ** 2 * delta / a**5
    friction = -(3 / a + dH_da / H_val) * d_delta_da
    return [d_delta_da, source + friction]


def fs8_theory(lk: Likelihood, theta, rtol=1e-6, atol=1e-8, method="RK45"):
    """fs8/fs8.py:84-98: scipy's adaptive integrator with the reference's tolerances, PCHIP of delta' at the data points."""
    from scipy.integrate import solve_ivp

    a_span = lk.fs8_a_span
    sol = solve_ivp(growth_ODE, t_span=(a_span[0], a_span[-1]), y0=(a_span[0], 1.0), t_eval=a_span, rtol=rtol, atol=atol,
                    method=method, args=(lk, np.asarray(theta, dtype=np.float64)))
    delta, d_delta_da = sol.y
    a = 1 / (1.0 + lk.fs8_z)
    return (lk.s8.get(theta) / delta[-1]) * a * interp_pchip(a, a_span, d_delta_da)


def chi2_fs8(lk: Likelihood, theta, tables=None, **ivp) -> float:
    """fs8/fs8.py:111-120: Alcock-Paczynski factor q = H D_M / (H D_M)_fid, chi2 = f_err^2 delta C^-1 delta."""
    cum_dm, dh_grid = tables if tables is not None else dm_grid(lk, theta)
    q = H_z(lk, lk.fs8_z, theta) * interp_hermite(lk.fs8_z, lk.z_grid, cum_dm, dh_grid) / lk.fs8_fid
    delta = lk.fs8_val - fs8_theory(lk, theta, **ivp) / q
    return float(lk.fs8err.get(theta) ** 2 * (delta @ lk.fs8_inv_cov @ delta))


def chi2_blocks(lk: Likelihood, theta):
    """(sn, bao, cmb) blocks + Gaussian chi^2 terms; bao/desi_cmb_des5y.py:138-141."""
    theta = np.asarray(theta, dtype=np.float64)
    tables = dm_grid(lk, theta) if (lk.z_cmb is not None or lk.bao_z is not None) else None  # (the growth block builds its own)
    sn = bao = cmb = 0.0
    if lk.z_cmb is not None:
        *_, delta = sn_parts(lk, theta, tables)
        sn = solve_triangular_chi2(lk.chol, delta)
    if lk.bao_z is not None:
        bao = chi2_bao(lk, theta, tables)
    if lk.cmb_mode:
        cmb = chi2_cmb(lk, theta)
    return sn, bao, cmb


def chi_squared(lk: Likelihood, theta) -> float:
    """sn/pantheon.py:57-61; joint: bao/desi_cmb_des5y.py:138-141."""
    theta = np.asarray(theta, dtype=np.float64)
    sn, bao, cmb = chi2_blocks(lk, theta)
    total = cmb + bao + sn
    if lk.cc_z is not None:
        total += chi2_cc(lk, theta)
    if lk.fs8_z is not None:
        total += chi2_fs8(lk, theta)
    for idx, mean, sigma in lk.chi2_gauss:
        total += (theta[idx] - mean) ** 2 / sigma**2
    return total


def log_likelihood(lk: Likelihood, theta) -> float:
    theta = np.asarray(theta, dtype=np.float64)
    if lk.cpl_wall and lk.w0.get(theta) + lk.wa.get(theta) >= 0.0:
        return -1e8  # bao/desi_fs_lya_cmb.py:118-121
    ll = -0.5 * chi_squared(lk, theta) + lk.logl_const
    if lk.fs8_z is not None:  # fs8/fs8.py:123-125: -0.5 (chi2 - 2 N ln f_err)
        ll += len(lk.fs8_z) * np.log(lk.fs8err.get(theta))
    if lk.cc_z is not None:  # bao/desi_union3_cc_theta_star.py:135-139
        n_cc = len(lk.cc_z)
        sign = 2 if lk.cc_f_inverse else -2  # ohd/cc_pantheon.py:92
        ll -= 0.5 * (n_cc * np.log(2 * np.pi) + lk.cc_logdet + sign * n_cc * np.log(lk.fcc.get(theta)))
    return ll


def log_prior(lk: Likelihood, theta) -> float:
    """sn/pantheon.py:80-85: strict box, flat normalisation, optional Gaussian terms."""
    theta = np.asarray(theta, dtype=np.float64)
    b = lk.bounds
    if b is not None:
        if not np.all((b[:, 0] < theta) & (theta < b[:, 1])):
            return -np.inf
        lp = -np.sum(np.log(b[:, 1] - b[:, 0])) if lk.prior_normalised else 0.0
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
    return lp + log_likelihood(lk, theta)


def log_probs_vectorized(lk: Likelihood, batch) -> np.ndarray:
    """bao/desi.py:100-106 (row loop); float64 out (superset of the reference's float32)."""
    return np.array([log_probability(lk, row) for row in np.atleast_2d(batch)])
