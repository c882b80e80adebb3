"""GPU (-m gpu): seeded random likelihood shapes against the numpy oracle -- block combinations, sizes at and around the tile /
panel / grid boundaries (N = 1 .. 700, G = 6 .. 4500 incl. the LDS chunk path above 4096 nodes, W ragged), both solve kernels.
Edge cases the reference's own scripts never reach but the C-ABI accepts."""
import os

import numpy as np
import pytest

from oracle import oracle_np as onp
from test_oracle_golden import _cmbdata, _phys

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope="module")
def gpu(pkg):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    return pkg


def _sn(rng, n, z_hi):
    z = np.sort(rng.uniform(0.01, z_hi, n))
    zh = z * (1 + 1e-3 * rng.standard_normal(n))
    sig = rng.uniform(0.1, 0.3, n)
    A = 0.02 * rng.standard_normal((n, min(n, 12)))
    cov = np.diag(sig**2) + A @ A.T
    obs = 25 + 5 * np.log10((1 + zh) * 4283.0 * z * (1 + 0.4 * z)) - 19.3 + 0.15 * rng.standard_normal(n)
    return z, zh, obs, np.linalg.cholesky(cov)


# CF_TEST_RANDOM_SHAPES=<n>: soak with more seeds (the default 24 keep the suite short)
@pytest.mark.parametrize("seed", range(int(os.environ.get("CF_TEST_RANDOM_SHAPES", "24"))))
def test_random_shape(gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    P, S = gpu.Param, onp.Slot
    physical = seed % 3 == 2
    fde = int(rng.integers(0, 4))
    n_sn = int(rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 200, 257, 700]))
    n_grid = int(rng.choice([6, 7, 64, 513, 4000, 4096, 4100, 4500]))
    n_bao = int(rng.choice([0, 1, 5, 14]))
    use_cc = bool(rng.integers(0, 2)) and not physical
    if n_sn == 0 and n_bao == 0 and not use_cc:
        n_bao = 3
    z_hi = float(rng.uniform(0.3, 2.3))
    solve = ["auto", "blocked"][seed % 2]
    comp = _cmbdata("PLANCK_ACT")
    names = ["offset", "H0"] + (["obh2", "och2"] if physical else ["Om"]) + ["v"] + (["w0"] if fde else []) + (["wa"] if fde == 3 else [])
    if not physical:
        names.append("rd")
    if use_cc:
        names.append("fcc")
    ndim = len(names)
    idx = {n: i for i, n in enumerate(names)}
    box = dict(offset=(-19.6, -19.0), H0=(60, 80), Om=(0.15, 0.5), obh2=(0.02, 0.025), och2=(0.09, 0.14), v=(-3, 3), w0=(-1.3, -0.6),
               wa=(-1.0, 0.2), rd=(130, 160), fcc=(0.5, 2.0))
    kw_g, kw_o = dict(params={n: P(i) for n, i in idx.items()}), {n: S(i) for n, i in idx.items()}
    z_max = z_hi + 0.1
    eng_kw = dict(ndim=ndim, z_max=z_max, n_grid=n_grid, fde=fde, ez_model=int(physical), solve_mode=gpu.engine.solve_mode_of(solve))
    olk_kw = dict(ndim=ndim, z_max=z_max, n_grid=n_grid, fde=fde, ez_model=int(physical))
    if physical:
        eng_kw["physical"] = {k: comp[k] for k in ("or_h2", "omnu_h2", "o_gamma_h2", "nu_m0", "nu_rho0", "nu_qs_sq", "nu_ws")}
        olk_kw.update(_phys(comp))
    if n_sn:
        z, zh, obs, chol = _sn(rng, n_sn, z_hi)
        step = None if seed % 4 else rng.uniform(-1, 1, n_sn)  # general weights every fourth case
        eng_kw["sn"] = dict(z_cmb=z, z_hel=zh, obs=obs, chol=chol, z_turn=0.15, step=step)
        olk_kw.update(z_cmb=z, z_hel=zh, obs=obs, chol=chol, z_turn=0.15, step=step)
    if n_bao:
        bz = np.sort(rng.uniform(0.05, z_hi, n_bao))
        if seed % 5 == 0:
            bz[0], bz[-1] = 1e-3, z_max  # a datum in the first interval and one ON the last grid node
        qty = rng.integers(0, 4, n_bao).astype(np.int32)
        M = rng.standard_normal((n_bao, n_bao))
        inv = M @ M.T + n_bao * np.eye(n_bao)
        val = rng.uniform(5, 30, n_bao)
        dh_exact = bool(rng.integers(0, 2)) or n_grid < 8
        bao = dict(z=bz, val=val, qty=qty, inv_cov=inv, dh_exact=dh_exact)
        olk_kw.update(bao_z=bz, bao_val=val, bao_qty=qty, bao_inv_cov=inv, bao_dh_exact=dh_exact)
        if physical:
            bao["rd_fit"] = comp["rd_fit"]
            olk_kw["rd_fit"] = comp["rd_fit"]
        eng_kw["bao"] = bao
    if physical and seed % 2 == 0:
        eng_kw["cmb"] = dict(mode=1, prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"])
        olk_kw.update(cmb_mode=1, cmb_prior=comp["cmb_prior"], cmb_inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"])
    if use_cc:
        n_cc = int(rng.integers(1, 33))
        cz = np.sort(rng.uniform(0.05, 2.0, n_cc))
        ch = 70 * np.sqrt(0.3 * (1 + cz) ** 3 + 0.7) + 5 * rng.standard_normal(n_cc)
        ccov = np.diag(rng.uniform(5, 20, n_cc) ** 2)
        eng_kw["cc"] = dict(z=cz, h=ch, inv_cov=np.linalg.inv(ccov), logdet=np.linalg.slogdet(ccov)[1])
        olk_kw.update(cc_z=cz, cc_h=ch, cc_inv_cov=np.linalg.inv(ccov), cc_logdet=np.linalg.slogdet(ccov)[1])
    bounds = np.array([box[n] for n in names], dtype=float)
    eng = gpu.LikelihoodEngine(bounds=bounds, **kw_g, **eng_kw)
    olk = onp.Likelihood(bounds=bounds, **kw_o, **olk_kw)
    W = int(rng.choice([1, 2, 31, 33, 100]))
    theta = gpu.synthetic.walkers(bounds, W, seed=seed)
    if W > 2:
        theta[1, 1] = 95.0  # one walker outside the box
    got_c, got_p = eng.chi_squared(theta), eng.log_probability(theta)
    with np.errstate(all="ignore"):
        want_c = np.array([onp.chi_squared(olk, t) for t in theta])
        want_p = np.array([onp.log_probability(olk, t) for t in theta])
    fin = np.isfinite(want_p)
    np.testing.assert_allclose(got_c, want_c, rtol=RTOL, err_msg=f"seed {seed}: {names}, N={n_sn}, G={n_grid}, B={n_bao}, physical={physical}, fde={fde}")
    np.testing.assert_allclose(got_p[fin], want_p[fin], rtol=RTOL)
    assert np.all(got_p[~fin] == -np.inf)
    # the same walkers inside a batch large enough for the throughput kernels (solve: 32-walker panels, residuals in walker rows;
    # per-walker kernel: one workgroup per walker; small blocks: sixteen lanes per walker): a walker's result must be the same BITS
    big = np.concatenate([theta, gpu.synthetic.walkers(bounds, 2200 - W, seed=seed + 1000)])
    np.testing.assert_array_equal(eng.chi_squared(big)[:W], got_c)
    eng.close()
