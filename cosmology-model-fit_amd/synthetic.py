"""
Synthetic Pantheon+-shaped supernova data: the real Pantheon+/DES covariances are
not in the reference snapshot and there is no network, so bench.py, smoke() and the tests use
this seeded recipe.  Plain numpy, host side, run once before an engine is created.
"""
import numpy as np

C_KM_S = 299792.458
THETA_TRUE = np.array([-19.35, 70.4, 0.315, 0.0])  # M, H0, Om, v


def _mu_true(z_cmb, z_hel, H0, Om):
    zz = np.linspace(0.0, z_cmb.max() * 1.001, 20001)
    dh = C_KM_S / (H0 * np.sqrt(Om * (1 + zz) ** 3 + (1 - Om)))
    dm = np.concatenate([[0.0], np.cumsum(0.5 * (dh[1:] + dh[:-1]) * np.diff(zz))])
    return 25.0 + 5 * np.log10((1 + z_hel) * np.interp(z_cmb, zz, dm))


def pantheon_like(n_sn=1701, seed=0, rank=40):
    """Returns dict(z_cmb, z_hel, obs, cov, chol, z_max): 45 % of the SNe in [0.01, 0.15], the rest out to
    2.26; C = diag(sigma^2) + A A^T with sigma in [0.1, 0.3] mag and A = 0.01 N(0,1)[N, rank]."""
    rng = np.random.default_rng(seed)
    n_lo = int(round(0.45 * n_sn))
    z = np.sort(np.concatenate([rng.uniform(0.01, 0.15, n_lo),
                                np.exp(rng.uniform(np.log(0.15), np.log(2.26), n_sn - n_lo))]))
    z_hel = z * (1 + 1e-3 * rng.standard_normal(n_sn))
    sigma = rng.uniform(0.1, 0.3, n_sn)
    A = 0.01 * rng.standard_normal((n_sn, rank))
    cov = np.diag(sigma**2) + A @ A.T
    chol = np.linalg.cholesky(cov)
    M, H0, Om, _ = THETA_TRUE
    obs = _mu_true(z, z_hel, H0, Om) + M + chol @ rng.standard_normal(n_sn)
    return dict(z_cmb=z, z_hel=z_hel, obs=obs, cov=cov, chol=chol, z_max=float(z.max() + 0.1))


def walkers(bounds, W, seed=0):
    """W walker positions uniform inside the prior box (all in-prior, none short-circuits)."""
    b = np.asarray(bounds, dtype=np.float64)
    rng = np.random.default_rng(seed + 1)
    lo, hi = b[:, 0], b[:, 1]
    eps = 1e-9 * (hi - lo)
    return rng.uniform(lo + eps, hi - eps, size=(W, len(lo)))
