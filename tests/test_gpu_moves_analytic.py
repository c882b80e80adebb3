"""
GPU (-m gpu): the three proposal moves of the device-resident sampler, pinned on the OUTPUTS OF THE HIP KERNELS (cf_ens_propose,
cf_ens_kde_prepare) against what emcee's published moves are defined to do -- not against the builder's own tensor statement
(oracle/moves_torch.py), and without emcee, which is not installed here.  The reference configures
them at sn/pantheon.py:114-117: KDEMove(bw_method="silverman") 30 %, DEMove() 70 %; StretchMove is emcee's default.

  StretchMove  y = c + z (x - c), c a walker of the complementary set drawn uniformly, z ~ g(z) proportional to 1 / sqrt(z) on
               [1 / a, a] (a = 2), log Hastings factor (ndim - 1) ln z            (Goodman & Weare 2010; emcee.moves.StretchMove)
  DEMove       y = x + gamma (c_j - c_k), (j, k) an ordered pair of DIFFERENT complementary walkers drawn uniformly,
               gamma = gamma0 (1 + sigma n), gamma0 = 2.38 / sqrt(2 ndim), sigma = 1e-5, n ~ N(0, 1), factor 0     (emcee.moves.DEMove)
  KDEMove      y ~ scipy.stats.gaussian_kde(complement, bw_method="silverman"), factor log kde(x) - log kde(y)      (emcee.moves.KDEMove)
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pkg):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    return pkg


class Kernels:
    """cf_ens_active_set / cf_ens_kde_prepare / cf_ens_propose on one set of positions; everything else is numpy / scipy."""

    def __init__(self, gpu, pos, n_splits):
        import torch

        self.torch, self.gpu, self.E = torch, gpu, gpu.ensemble
        self.lib, self.L = gpu.lib(), gpu._lib
        self.dev = torch.device("cuda:0")
        self.pos_host = np.ascontiguousarray(pos)
        self.pos = torch.from_numpy(self.pos_host).to(self.dev)
        self.n_total, self.ndim = pos.shape
        self.S = n_splits
        self.stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.kde_params = torch.empty(2 * self.ndim * self.ndim + 1, dtype=torch.float64, device=self.dev)
        self.kde_wc = torch.empty((self.n_total, self.ndim), dtype=torch.float64, device=self.dev)

    def propose(self, kind, step, split, a=2.0, de_sigma=1e-5, seed=11):
        """(ids of the active walkers, their positions, the complementary set in ascending walker order, y, log factor)."""
        torch, lib, L, E = self.torch, self.lib, self.L, self.E
        split_key = E.stream_key(seed, step, 0, E._SPLIT_STREAM)  # the per-step re-drawn splits of emcee's RedBlueMove
        n = int(lib.cf_ens_active_count(split_key, self.S, split, 0, self.n_total))
        ids = torch.empty(n, dtype=torch.int64, device=self.dev)
        idx = torch.empty(n, dtype=torch.int64, device=self.dev)
        L.check(lib.cf_ens_active_set(split_key, self.S, split, 0, self.n_total, ids.data_ptr(), idx.data_ptr(), self.stream))
        if kind == 2:
            L.check(lib.cf_ens_kde_prepare(self.pos.data_ptr(), self.n_total, self.ndim, self.S, split, split_key,
                                           self.kde_params.data_ptr(), self.kde_wc.data_ptr(), self.stream))
        y = torch.empty((n, self.ndim), dtype=torch.float64, device=self.dev)
        lf = torch.empty(n, dtype=torch.float64, device=self.dev)
        L.check(lib.cf_ens_propose(kind, self.pos.data_ptr(), self.n_total, self.ndim, self.S, split, split_key, ids.data_ptr(), n,
                                   E.stream_key(seed, step, split), a, de_sigma, self.kde_params.data_ptr(), self.kde_wc.data_ptr(),
                                   y.data_ptr(), lf.data_ptr(), self.stream))
        torch.cuda.synchronize()
        ids_h = ids.cpu().numpy()
        active = np.zeros(self.n_total, dtype=bool)
        active[ids_h] = True
        assert n == active.sum() and lib.cf_ens_comp_count(split_key, self.S, split, self.n_total) == self.n_total - n
        return ids_h, self.pos_host[ids_h], self.pos_host[~active], y.cpu().numpy(), lf.cpu().numpy()


def _uniform_counts_ok(counts, n_draws):
    """Pearson chi^2 of `counts` against the uniform distribution over len(counts) cells, 5 sigma of its own spread."""
    k = len(counts)
    chi2 = ((counts - n_draws / k) ** 2 / (n_draws / k)).sum()
    return abs(chi2 - (k - 1)) < 5 * math.sqrt(2 * (k - 1)), chi2


def test_stretch_move_of_the_kernels_is_goodman_weare(gpu):
    from scipy import stats

    ndim, a = 4, 2.0
    rng = np.random.default_rng(0)
    K = Kernels(gpu, rng.standard_normal((1200, ndim)) * np.array([0.5, 1.0, 2.0, 4.0]) + 3.0, 2)
    zs, partner_counts, n_draws = [], np.zeros(0), 0
    for step in range(100):
        for split in (0, 1):
            ids, x, comp, y, lf = K.propose(0, step, split, a=a)
            z = np.exp(lf / (ndim - 1))  # log factor = (ndim - 1) ln z: the z^(ndim - 1) of the acceptance ratio
            assert np.all(z >= 1 / a - 1e-12) and np.all(z <= a + 1e-12)
            c = (y - z[:, None] * x) / (1.0 - z)[:, None]  # y = c + z (x - c)
            d2 = ((c[:, None, :] - comp[None, :, :]) ** 2).sum(axis=2)
            j = d2.argmin(axis=1)
            scale = np.abs(x).max() / np.minimum(np.abs(1 - z), 1.0)
            assert np.all(np.sqrt(d2[np.arange(len(j)), j]) < 1e-9 * scale), "the partner is a walker of the complementary set"
            if len(partner_counts) != len(comp):
                partner_counts = np.zeros(len(comp))
            partner_counts += np.bincount(j, minlength=len(comp))
            n_draws += len(j)
            zs.append(z)
    z = np.concatenate(zs)
    assert z.size == 100 * 1200
    cdf = lambda t: (np.sqrt(t) - 1 / math.sqrt(a)) / (math.sqrt(a) - 1 / math.sqrt(a))  # of g(z) = 1 / sqrt(z) / norm on [1/a, a]
    ks = stats.kstest(z, cdf)
    assert ks.statistic < 1.95 / math.sqrt(z.size), f"z is not drawn from g(z): KS D = {ks.statistic:.2e} on {z.size} draws"  # p > 1e-3
    mean_g, var_g = (a + 1 + 1 / a) / 3, (a * a + a + 1 + 1 / a + 1 / (a * a)) / 5 - ((a + 1 + 1 / a) / 3) ** 2
    assert abs(z.mean() - mean_g) < 5 * math.sqrt(var_g / z.size)
    ok, chi2 = _uniform_counts_ok(partner_counts, n_draws)
    assert ok, f"partners are not uniform over the complementary set: chi2 = {chi2:.0f} for {len(partner_counts)} cells"


@pytest.mark.parametrize("ndim,n_splits", [(4, 3), (6, 3), (4, 2)])
def test_de_move_of_the_kernels_is_emcees_demove(gpu, ndim, n_splits):
    from scipy import stats

    sigma, g0 = 1e-5, 2.38 / math.sqrt(2 * ndim)  # emcee.moves.DEMove: sigma = 1.0e-5, gamma0 = 2.38 / sqrt(2 ndim)
    rng = np.random.default_rng(ndim)
    K = Kernels(gpu, rng.standard_normal((60, ndim)) * np.linspace(0.5, 3.0, ndim) - 1.0, n_splits)
    gam, pair_counts, n_draws = [], None, 0
    for step in range(400):
        for split in range(n_splits):
            ids, x, comp, y, lf = K.propose(1, step, split, de_sigma=sigma)
            assert np.all(lf == 0.0), "the DE proposal is symmetric: log Hastings factor 0"
            nc = len(comp)
            assert nc == 60 - len(ids) and (n_splits == 2 or abs(len(ids) - 20) <= 1)  # three splits: thirds of the ensemble
            diff = comp[:, None, :] - comp[None, :, :]                      # [j, k] = c_j - c_k
            nrm2 = (diff ** 2).sum(axis=2) + np.eye(nc)                     # (the diagonal is excluded below)
            d = y - x
            g = np.einsum("nd,jkd->njk", d, diff) / nrm2                    # least-squares gamma for every ordered pair
            res = ((d[:, None, None, :] - g[..., None] * diff[None]) ** 2).sum(axis=3) + np.eye(nc)[None] * 1e300
            flat = res.reshape(len(d), -1).argmin(axis=1)
            j, k = flat // nc, flat % nc
            best = res.reshape(len(d), -1)[np.arange(len(d)), flat]
            assert np.all(j != k) and np.all(np.sqrt(best) < 1e-10 * np.sqrt((d ** 2).sum(axis=1))), "y - x is gamma (c_j - c_k), j != k"
            gsel = g[np.arange(len(d)), j, k]
            # (j, k) with gamma and (k, j) with -gamma fit equally well; emcee's gamma0 (1 + sigma n) is positive, which fixes the order
            j, k, gsel = np.where(gsel < 0, k, j), np.where(gsel < 0, j, k), np.abs(gsel)
            gam.append(gsel)
            if pair_counts is None or pair_counts.shape != (nc, nc):
                pair_counts = np.zeros((nc, nc))
            np.add.at(pair_counts, (j, k), 1)
            n_draws += len(d)
    gam = np.concatenate(gam)
    assert np.all(gam > 0) and gam.size >= 400 * 58
    n = (gam / g0 - 1.0) / sigma  # gamma = gamma0 (1 + sigma n), n ~ N(0, 1)
    assert abs(gam.mean() / g0 - 1.0) < 5 * sigma / math.sqrt(gam.size), f"gamma0 = {gam.mean():.6f}, emcee's is {g0:.6f}"
    assert abs(n.std() - 1.0) < 5 / math.sqrt(2 * gam.size), f"sigma = {n.std() * sigma:.3e}, emcee's is {sigma:.1e}"
    assert stats.kstest(n, "norm").statistic < 1.95 / math.sqrt(n.size)
    off = ~np.eye(pair_counts.shape[0], dtype=bool)
    ok, chi2 = _uniform_counts_ok(pair_counts[off], n_draws)
    assert ok and pair_counts[~off].sum() == 0, f"ordered pairs are not uniform: chi2 = {chi2:.0f} for {off.sum()} cells"
    asym = pair_counts - pair_counts.T  # ORDERED pairs: (j, k) and (k, j) are both drawn, equally often
    assert abs(asym[np.triu_indices_from(asym, 1)]).max() < 6 * math.sqrt(2 * n_draws / off.sum()) + 6


@pytest.mark.parametrize("ndim,n_total", [(4, 300), (2, 64), (6, 150)])
def test_kde_move_of_the_kernels_is_scipys_silverman_kde(gpu, ndim, n_total):
    from scipy import stats

    rng = np.random.default_rng(10 + ndim)
    A = rng.standard_normal((ndim, ndim)) * 0.4 + np.eye(ndim)  # correlated cloud
    K = Kernels(gpu, rng.standard_normal((n_total, ndim)) @ A.T + np.arange(ndim), 2)
    ys, n_prop = [], 0
    comp0 = None
    for step in range(30 if n_total >= 150 else 150):
        ids, x, comp, y, lf = K.propose(2, step, 0)
        kde = stats.gaussian_kde(comp.T, bw_method="silverman")
        nc = len(comp)
        assert kde.factor == pytest.approx((nc * (ndim + 2) / 4.0) ** (-1.0 / (ndim + 4)), rel=1e-14)  # scipy's silverman_factor
        # emcee.moves.KDEMove: factor = kde.logpdf(x) - kde.logpdf(y), with scipy's own bandwidth, covariance (ddof = 1) and density
        want = kde.logpdf(x.T) - kde.logpdf(y.T)
        np.testing.assert_allclose(lf, want, rtol=1e-9, atol=1e-9)
        if step == 0:
            comp0 = comp
        if comp.shape == comp0.shape and np.array_equal(comp, comp0):
            ys.append(y)
        # every proposal lies where the KDE has mass: within 8 bandwidth-sigmas (Mahalanobis) of some complementary walker
        Li = np.linalg.inv(np.linalg.cholesky(kde.covariance))
        wy, wc = y @ Li.T, comp @ Li.T
        assert np.sqrt(((wy[:, None, :] - wc[None, :, :]) ** 2).sum(axis=2).min(axis=1)).max() < 8.0
        n_prop += len(y)
    # the proposal DISTRIBUTION: moments of draws from a Gaussian KDE -- mean = mean of the centres, covariance = population
    # covariance of the centres + h^2 x their sample covariance; the per-step re-drawn splits change the complementary set, so
    # instead the draws of every step are whitened with that step's own KDE moments
    zs = []
    for step in range(200):
        ids, x, comp, y, lf = K.propose(2, step, 1)
        kde = stats.gaussian_kde(comp.T, bw_method="silverman")
        cov = np.cov(comp.T, bias=True).reshape(ndim, ndim) + kde.covariance.reshape(ndim, ndim)
        zs.append((y - comp.mean(axis=0)) @ np.linalg.inv(np.linalg.cholesky(cov)).T)
    z = np.concatenate(zs)
    n = len(z)
    assert np.abs(z.mean(axis=0)).max() < 5 / math.sqrt(n)
    c = z.T @ z / n
    assert np.abs(c - np.eye(ndim)).max() < 6 * math.sqrt(3.0 / n) + 0.02  # (a KDE is not Gaussian: fourth moments above 3 widen this)
