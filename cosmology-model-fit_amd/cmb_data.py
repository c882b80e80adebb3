"""
Constants (data, not code) of the reference's compressed-CMB modules and of its 5-node massive-neutrino
density, needed to fill the CMB / radiation fields of ``cf_desc``.

Sources: cmb/data_planck_act_compression.py:15-50,90,106 (Planck + ACT DR6 (R, l_A, omega_b) priors,
arXiv:2503.14452), cmb/data_early_lcdm_compression.py:13-48,88,104 (early-LCDM (theta*, omega_b, omega_m),
arXiv:2302.12911), fitting-formula shapes of arXiv:2106.00428, nu_evolution.py:10-28 (5-node coefficients).
Every derived number is re-computed here with the reference's own expressions and pinned against golden
values in tests/test_oracle_golden.py (z_star, r_drag, Omnu_z, Or_h2, Omnu_h2, m0, rho0, qs^2).
"""
import numpy as np

K_B = 8.617333262e-5  # eV/K
TCMB = 2.7255  # K
O_GAMMA_H2 = 2.472975328714087e-05
N_EFF = 3.044
MNU_TOT = 0.06  # eV

# nu_evolution.py:10-20: q_i(m0) = a + b / (m0**d + c), weights
_NU_COEFFS = (
    (0.51957626, -0.32971882, 61.81645189, 1.63914879),
    (1.44003028, +0.18098045, 55.20830625, 1.62412241),
    (2.98731126, -0.15154978, 38.50221716, 1.54532306),
    (5.51951238, +0.27573416, 27.23306000, 1.54350910),
    (9.82330637, -1.14831159, 14.84585003, 1.55585284),
)
NU_WEIGHTS = np.array([0.0380051, 0.262676, 0.46542, 0.217161, 0.0167379])

# z_star exponents / amplitudes common to both compressions (cmb/data_planck_act_compression.py:94-99)
ZSTAR_CONSTS = (-0.7316314841257655, 391.6723594873167, 0.9368102670600895, -0.35300106475765136,
                937.4224935298015, 0.0192950634264157, -0.04285000485853785)
# r_drag a1..a9 (cmb/data_planck_act_compression.py:111-119)
RDRAG_A = (0.00257366, 0.05032, 0.013, 0.7720642, 0.24346362, 0.00641072, 0.5350899, 32.7525, 0.315473)


def _neutrino(omnu_denominator, N_EFF=N_EFF):
    T_nu0 = (4 / 11) ** (1 / 3) * (N_EFF / 3) ** (1 / 4) * TCMB
    m0 = MNU_TOT / (T_nu0 * K_B)
    qs = np.array([a + b / (m0**d + c) for a, b, c, d in _NU_COEFFS], dtype=np.float64)
    rho0 = 0.0
    for i in range(5):
        rho0 += NU_WEIGHTS[i] * np.sqrt(qs[i] ** 2 + m0**2)
    omnu_h2 = MNU_TOT / (omnu_denominator / (N_EFF / 3.0) ** 0.75)
    neff = 2 * N_EFF / 3
    or_h2 = O_GAMMA_H2 * (1 + neff * (7 / 8) * (4 / 11) ** (4 / 3))
    return dict(nu_m0=m0, nu_rho0=rho0, nu_qs_sq=qs**2, nu_ws=NU_WEIGHTS.copy(), omnu_h2=omnu_h2, or_h2=or_h2,
                o_gamma_h2=O_GAMMA_H2)


def _compression(priors, covariance, omnu_den, zstar_sbm, rdrag_bm, mode, n_eff=N_EFF):
    d = _neutrino(omnu_den, n_eff)
    d.update(cmb_prior=np.array(priors), cmb_cov=np.array(covariance), cmb_inv_cov=np.linalg.inv(np.array(covariance)),
             zstar_fit=tuple(zstar_sbm), rd_fit=tuple(rdrag_bm) + RDRAG_A, cmb_mode=mode)
    return d


# cmb_mode: 1 = (R, l_A, omega_b), 3 = (theta*, omega_b, omega_m)   (include/cosmofit.h cf_cmb_mode)
PLANCK_ACT = _compression(
    [1.74795802, 301.803306, 0.0224962530],
    [[1.54911112e-05, 1.03997132e-04, -2.10953275e-07],
     [1.03997132e-04, 5.43880523e-03, -1.53612827e-06],
     [-2.10953275e-07, -1.53612827e-06, 1.23574770e-08]],
    94.0641, (0.70130133, 1.00839438, 1.02468387, 1.18438972), (0.99625075, 1.00593295), 1)

EARLY_LCDM = _compression(
    [0.010410274, 0.02223, 0.14208],
    1e-9 * np.array([[0.00662099420, 0.124442058, -1.19287532],
                     [0.124442058, 21.3441666, -94.0008323],
                     [-1.19287532, -94.0008323, 1488.41714]]),
    94.07, (0.75717491, 1.00737989, 1.02737182, 1.20432292), (1.00140649, 1.00072621), 3)

# cmb/data_planck_compression.py:12-33,87,103 (Planck 2018 (R, l_A, omega_b), N_eff = 3.046)
PLANCK = _compression(
    [1.75063846, 301.760701, 0.0223597502],
    [[2.09107356e-05, 1.78419597e-04, -4.46283183e-07],
     [1.78419597e-04, 7.81249750e-03, -4.24834772e-06],
     [-4.46283183e-07, -4.24834772e-06, 2.21402189e-08]],
    94.07, (0.73491615, 1.00820929, 1.01709662, 1.17030559), (1.00078696, 1.00128548), 1, n_eff=3.046)

BBN_SCHONEBERG = (0.02218, 0.00055)  # omega_b mean, sigma: y2024BBN/prior_lcdm_schoneberg.py:2-3


# Derived parameters of the post-fit blocks (gd_samples.addDerived(cmb.z_star(...)), bao/desi_cmb_union3.py:181-192): the
# fitting formulae with this compression's constants, vectorised over the posterior samples on the host.
def z_star(comp, wb, wm):
    """Redshift of photon decoupling, arXiv:2106.00428 eq. A-4 (cmb/data_planck_act_compression.py:86-99)."""
    s1, s2, b, m = comp["zstar_fit"]
    e0, a1, e1, e2, a2, e3, e4 = ZSTAR_CONSTS
    wb, wm = np.asarray(wb, float) ** b, np.asarray(wm, float) ** m
    return wm**e0 + s1 * a1 * wb**e1 * wm**e2 + s2 * a2 * wm**e3 * wb**e4


def r_drag(comp, wb, wm):
    """Sound horizon at the drag epoch in Mpc, arXiv:2106.00428 eq. 8 (cmb/data_planck_act_compression.py:102-124)."""
    b, m, a1, a2, a3, a4, a5, a6, a7, a8, a9 = comp["rd_fit"]
    wb, wm = np.asarray(wb, float) ** b, np.asarray(wm, float) ** m
    return 1.0 / (a1 * wb**a2 + a3 * wb**a4 * wm**a5 + a6 * wm**a7) - a8 / wm**a9
