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


def hard_cov(z, sigma, seed=0):
    """A seeded covariance with the structure of the Pantheon+ STAT+SYS matrix that the reference snapshot lacks
    (.MISSING_LARGE_BLOBS), for a z-sorted sample with statistical errors `sigma`:

      * duplicated supernovae: ~18 % of the entries are 2-4 observations of one SN by different surveys, adjacent in
        z, correlated with rho in [0.99, 0.99995];
      * one fully coherent 0.05 mag systematic (a calibration offset shared by every SN);
      * seven survey blocks (contiguous in z with ragged edges) with their own 0.015-0.04 mag zero-point offsets;
      * four smooth-in-z systematics (0.01-0.03 mag, polynomial in log z);
      * 0.004 mag of low-rank noise so that no two rows are proportional.

    cond(C) is ~1e7 (tests/test_hard_cov.py prints it): the conditioning-sensitive case for chi^2 = ||L^-1 Delta||^2."""
    rng = np.random.default_rng(1000 + seed)
    z = np.asarray(z, dtype=np.float64)
    sigma = np.asarray(sigma, dtype=np.float64)
    n = z.size
    C = np.diag(sigma**2)
    # duplicated SNe: groups of adjacent entries sharing the intrinsic scatter
    i = 0
    while i < n - 4:
        if rng.uniform() < 0.07:
            k = int(rng.integers(2, 5))
            rho = 1.0 - 10 ** rng.uniform(-4.3, -2.0)
            for a in range(i, i + k):
                for b in range(i, i + k):
                    if a != b:
                        C[a, b] = rho * sigma[a] * sigma[b]
            i += k
        else:
            i += 1
    C += 0.05**2  # coherent systematic
    edges = np.sort(rng.choice(np.arange(50, n - 50), size=6, replace=False))
    survey = np.searchsorted(edges, np.arange(n) + rng.integers(-20, 21, n))  # ragged survey boundaries
    for s in range(7):
        m = (survey == s).astype(np.float64)
        C += rng.uniform(0.015, 0.04) ** 2 * np.outer(m, m)
    lz = np.log(z / z.min() + 1e-3)
    lz = (lz - lz.mean()) / lz.std()
    for p in range(1, 5):
        f = lz**p / np.max(np.abs(lz**p))
        C += rng.uniform(0.01, 0.03) ** 2 * np.outer(f, f)
    A = 0.004 * rng.standard_normal((n, 20))
    C += A @ A.T
    return 0.5 * (C + C.T)
