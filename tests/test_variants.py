"""
Parameterisation variants of the reference's scripts: a free sound horizon and the omega_m = Omega_m h^2
slot (bao/desi_omh2.py, bao/desi_des5y_rd.py), the linearised bulk-flow magnitude term (bao/desi_cmb_pantheon_H0trgb.py:102-106),
the direction-dependent peculiar velocity (sn/pantheon_dipole_xyz.py:50-60), BASELINE configs[2] as worded (w0waCDM on
bao/desi_cmb_des5y.py), and the reference's accessor signatures DM_z(params, z) / mu_theory(DM) / mu_corr(params, DM).

CPU: the numpy oracle against the fixtures the reference itself produced (tests/golden/generate_golden.py).
GPU (-m gpu): the mirrors through the C-ABI against the same fixtures, bar 1e-10.
"""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle_np as onp
from test_oracle_golden import _chol_of, _cmbdata, _phys

RTOL = 1e-10


# ---- oracle descriptions -----------------------------------------------------------------------------------------------
def lk_bao_desi_omh2(g):
    return onp.Likelihood(ndim=4, z_max=float(g["z_max"]), fde=onp.FDE_THAWING, om_mode=1, rd=onp.Slot(0), H0=onp.Slot(1),
                          Om=onp.Slot(2), w0=onp.Slot(3), bao_z=g["bao_z"], bao_val=g["bao_val"], bao_qty=g["bao_qty"],
                          bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True)


def lk_bao_desi_des5y_rd(g, chol):
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), offset=onp.Slot(0), rd=onp.Slot(1), H0=onp.Slot(2), Om=onp.Slot(3),
                          v=onp.Slot(4), z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], z_turn=0.10563, chol=chol,
                          bao_z=g["bao_z"], bao_val=g["bao_val"], bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"])


def lk_bao_desi_cmb_pantheon_H0trgb(g, chol):
    d = _cmbdata("PLANCK_ACT")
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, offset=onp.Slot(0), H0=onp.Slot(1),
                          obh2=onp.Slot(2), och2=onp.Slot(3), lin=onp.Slot(4), has_vstep=False,
                          lin_coef=100 * (5 / np.log(10)) / (onp.C_KM_S * g["z_cmb"]),
                          z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], chol=chol, bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True, rd_fit=d["rd_fit"], cmb_mode=1,
                          cmb_prior=d["cmb_prior"], cmb_inv_cov=d["cmb_inv_cov"], zstar_fit=d["zstar_fit"],
                          chi2_gauss=[(1, 70.39, 1.80)], **_phys(d))


def lk_sn_pantheon_dipole_xyz(g, chol):
    return onp.Likelihood(ndim=6, z_max=float(g["z_max"]), offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3),
                          v2=onp.Slot(4), v3=onp.Slot(5), z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], chol=chol,
                          step=g["weights"].astype(np.float64), dirs=g["dirs"])


def lk_bao_desi_cmb_des5y_cpl(g, base, chol):
    d = _cmbdata("PLANCK_ACT")
    return onp.Likelihood(ndim=7, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_CPL, offset=onp.Slot(0),
                          H0=onp.Slot(1), obh2=onp.Slot(2), och2=onp.Slot(3), v=onp.Slot(4), w0=onp.Slot(5), wa=onp.Slot(6),
                          z_cmb=base["z_cmb"], z_hel=base["z_hel"], obs=base["obs"], z_turn=0.10563, chol=chol,
                          bao_z=base["bao_z"], bao_val=base["bao_val"], bao_qty=base["bao_qty"], bao_inv_cov=base["bao_inv_cov"],
                          rd_fit=d["rd_fit"], cmb_mode=1, cmb_prior=d["cmb_prior"], cmb_inv_cov=d["cmb_inv_cov"],
                          zstar_fit=d["zstar_fit"], **_phys(d))


# ---- CPU: oracle vs the reference's fixtures ---------------------------------------------------------------------------------
def _check_oracle(lk, g, n=6):
    for k in list(range(min(n, len(g["thetas"]) - 2))) + [len(g["thetas"]) - 2, len(g["thetas"]) - 1]:
        assert onp.chi_squared(lk, g["thetas"][k]) == pytest.approx(g["chi2"][k], rel=RTOL)
        assert onp.log_likelihood(lk, g["thetas"][k]) == pytest.approx(g["logl"][k], rel=RTOL)


def test_oracle_bao_desi_omh2():
    g = golden("bao_desi_omh2")
    lk = lk_bao_desi_omh2(g)
    _check_oracle(lk, g, n=19)
    for k in range(4):
        np.testing.assert_allclose(onp.bao_theory(lk, g["thetas"][k]), g["theory"][k], rtol=1e-13)


def test_oracle_bao_desi_des5y_rd():
    g = golden("bao_desi_des5y_rd")
    _check_oracle(lk_bao_desi_des5y_rd(g, _chol_of(g)), g)


def test_oracle_bulk_flow_magnitude_term():
    g = golden("bao_desi_cmb_pantheon_H0trgb")
    lk = lk_bao_desi_cmb_pantheon_H0trgb(g, _chol_of(g))
    _check_oracle(lk, g, n=3)
    # the reference's apparent_mag = obs - residual
    for key, th in (("mag_0", g["thetas"][0]), ("mag_last", g["thetas"][-1])):
        *_, delta = onp.sn_parts(lk, th)
        np.testing.assert_allclose(g["obs"] - delta, g[key], rtol=0, atol=2e-13)


def test_oracle_dipole_xyz():
    g = golden("sn_pantheon_dipole_xyz")
    lk = lk_sn_pantheon_dipole_xyz(g, _chol_of(g))
    _check_oracle(lk, g)
    _, mucorr, _, _ = onp.sn_parts(lk, g["thetas"][0])
    np.testing.assert_allclose(mucorr, g["mucorr_0"], rtol=0, atol=1e-13)


def test_oracle_config3_as_worded_w0wa():
    g, base = golden("bao_desi_cmb_des5y_cpl"), golden("bao_desi_cmb_des5y")
    lk = lk_bao_desi_cmb_des5y_cpl(g, base, _chol_of(base))
    _check_oracle(lk, g, n=3)
    # w0 = -1, wa = 0 is the script as shipped: that row must equal the as-shipped fixture's value at the same theta
    assert np.array_equal(g["thetas"][-2][:5], base["thetas"][-2])
    assert g["chi2"][-2] == pytest.approx(base["chi2"][-2], rel=1e-13)


# ---- GPU: the mirrors through the C-ABI -----------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu(pkg):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    return pkg


def _bao_args(g):
    return g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"]


@pytest.mark.gpu
def test_gpu_desi_omh2_free_rd_and_physical_matter_density(gpu):
    g = golden("bao_desi_omh2")
    lk = gpu.likelihoods.DesiOmh2(*_bao_args(g))
    assert lk.z_max == float(g["z_max"])
    np.testing.assert_allclose(lk.chi_squared(g["thetas"]), g["chi2"], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"]), g["logl"], rtol=RTOL)
    for k in range(4):
        np.testing.assert_allclose(lk.bao_theory(g["thetas"][k]), g["theory"][k], rtol=1e-12)
    lk.engine.close()


@pytest.mark.gpu
def test_gpu_desi_des5y_rd(gpu):
    g = golden("bao_desi_des5y_rd")
    lk = gpu.likelihoods.DesiSnRd(g["z_cmb"], g["z_hel"], g["obs"], None, *_bao_args(g), chol=_chol_of(g))
    np.testing.assert_allclose(lk.chi_squared(g["thetas"]), g["chi2"], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"]), g["logl"], rtol=RTOL)
    for k in range(4):
        np.testing.assert_allclose(lk.bao_theory(g["thetas"][k]), g["theory"][k], rtol=1e-12)
    # the omega_m variant of the same script family (bao/desi_des5y_omh2.py) against the oracle
    lk2 = gpu.likelihoods.DesiSnRd(g["z_cmb"], g["z_hel"], g["obs"], None, *_bao_args(g), chol=_chol_of(g), omh2=True)
    olk = lk_bao_desi_des5y_rd(g, _chol_of(g))
    olk.om_mode = 1
    th = g["thetas"][:6].copy()
    th[:, 3] = th[:, 3] * (th[:, 2] / 100) ** 2
    np.testing.assert_allclose(lk2.chi_squared(th), [onp.chi_squared(olk, t) for t in th], rtol=RTOL)
    lk.engine.close()
    lk2.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("solve", ["auto", "blocked"])
def test_gpu_bulk_flow_magnitude_term(gpu, solve):
    g = golden("bao_desi_cmb_pantheon_H0trgb")
    lk = gpu.likelihoods.DesiCmbPantheonH0Trgb(g["z_cmb"], g["z_hel"], g["obs"], None, *_bao_args(g), chol=_chol_of(g), solve=solve)
    np.testing.assert_allclose(lk.chi_squared(g["thetas"]), g["chi2"], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"]), g["logl"], rtol=RTOL)
    # accessor path: residual = obs - apparent_mag of the reference
    d0 = lk.engine.parts(g["thetas"][:1])["delta"][0]
    np.testing.assert_allclose(g["obs"] - d0, g["mag_0"], rtol=0, atol=1e-12)
    lk.engine.close()


@pytest.mark.gpu
def test_gpu_dipole_velocity_vector(gpu):
    g = golden("sn_pantheon_dipole_xyz")
    lk = gpu.likelihoods.PantheonDipoleXyz(g["z_cmb"], g["z_hel"], g["obs"], None, g["dirs"], g["weights"], chol=_chol_of(g))
    np.testing.assert_allclose(lk.chi_squared(g["thetas"]), g["chi2"], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"]), g["logl"], rtol=RTOL)
    np.testing.assert_allclose(lk.engine.parts(g["thetas"][:1])["mu_corr"][0], g["mucorr_0"], rtol=0, atol=1e-13)
    # geometry helper = the reference's construction
    rng = np.random.default_rng(0)
    ra, dec, sid = rng.uniform(0, 360, 50), rng.uniform(-90, 90, 50), rng.choice([1, 4, 5, 150, 99], 50)
    dirs, w = lk.dipole_geometry(g["z_cmb"][:50], ra, dec, sid)
    np.testing.assert_allclose(np.linalg.norm(dirs, axis=1), 1.0, rtol=1e-15)
    assert set(np.unique(w[np.isin(sid, [4, 99])])) == {0.0}
    lk.engine.close()


@pytest.mark.gpu
def test_gpu_config3_as_worded_w0wa_full_size(gpu):
    """BASELINE configs[2] "w0waCDM": bao/desi_cmb_des5y.py with its commented CPL line active, N = 1820 + 14 BAO + CMB,
    against the fixture (reference functions with the patched H_z), then 4096 walkers against the numpy oracle on a subset
    and batch invariance."""
    g, base = golden("bao_desi_cmb_des5y_cpl"), golden("bao_desi_cmb_des5y")
    chol = _chol_of(base)
    lk = gpu.likelihoods.DesiCmbDes5y(base["z_cmb"], base["z_hel"], base["obs"], None, *_bao_args(base), chol=chol, fde="cpl")
    assert lk.ndim == 7
    np.testing.assert_allclose(lk.chi_squared(g["thetas"]), g["chi2"], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"]), g["logl"], rtol=RTOL)
    parts = lk.engine.parts(g["thetas"])
    np.testing.assert_allclose(parts["chi2_blocks"], g["chi2_parts"], rtol=1e-9)
    for k in range(4):
        np.testing.assert_allclose(parts["bao_theory"][k], g["theory"][k], rtol=1e-12)
        np.testing.assert_allclose(parts["cmb_vector"][k], g["cmb_dist"][k], rtol=1e-12)
    # the table accessor on a joint likelihood (BAO / CMB blocks present: their node copies must stay off in that launch)
    zq = np.array([0.0, 0.295, 0.51, 1.317, 2.33])
    cum, dh = onp.dm_grid(lk_bao_desi_cmb_des5y_cpl(g, base, chol), g["thetas"][0])
    np.testing.assert_allclose(lk.DM_z(zq, g["thetas"][0]), onp.interp_hermite(zq, np.linspace(0, lk.z_max, 4000), cum, dh), rtol=1e-12, atol=1e-9)
    box = np.array([(-0.5, 0.5), (60.0, 75.0), (0.010, 0.030), (0.01, 0.25), (-4.5, 4.5), (-3.0, -0.2), (-3.0, 0.1)])
    theta = gpu.synthetic.walkers(box, 4096, seed=3)
    got = lk.log_likelihood(theta)
    assert np.all(np.isfinite(got))
    olk = lk_bao_desi_cmb_des5y_cpl(g, base, chol)
    pick = np.arange(0, 4096, 128)
    want = np.array([onp.log_likelihood(olk, t) for t in theta[pick]])
    np.testing.assert_allclose(got[pick], want, rtol=RTOL)
    np.testing.assert_array_equal(lk.log_likelihood(theta[1000:3000]), got[1000:3000])
    lk.engine.close()


@pytest.mark.gpu
def test_gpu_reference_accessor_signatures(gpu, pantheon_golden):
    """sn/pantheon.py:34-54,152-155: DM_z(params, z), mu_theory(DM), mu_corr(params, DM) as the post-fit plot block calls them."""
    g = pantheon_golden
    lk = gpu.sn_pantheon.PantheonLikelihood(g["z_cmb"], g["z_hel"], g["obs"], chol=g["chol"], bounds=g["bounds"])
    for k in range(3):
        th = g["thetas"][k]
        DM = lk.DM_z(th, g["z_cmb"])
        np.testing.assert_allclose(DM, g[f"dm_{k}"], rtol=1e-13)
        np.testing.assert_allclose(lk.mu_theory(DM), g[f"muth_{k}"], rtol=1e-14)
        np.testing.assert_allclose(lk.mu_corr(th, DM), g[f"mucorr_{k}"], rtol=0, atol=1e-13)
        # the distance table itself = the reference's (cum_dm, dh_grid) sub-sampled in the fixture
        z_grid, cum, dh = lk.engine.distance_table(th)
        np.testing.assert_array_equal(z_grid[::250], g["z_grid_sub"])
        np.testing.assert_allclose(cum[0][::250], g[f"cum_sub_{k}"], rtol=1e-13)
        np.testing.assert_allclose(dh[0][::250], g[f"dh_sub_{k}"], rtol=1e-14)
    # arbitrary redshifts, outside the grid too (linear extrapolation, interpolator.py:87-92) vs the oracle
    zq = np.array([-0.01, 0.0, 1e-4, 0.5, 1.2345, float(g["z_max"]), float(g["z_max"]) + 0.2])
    olk = onp.Likelihood(ndim=4, z_max=float(g["z_max"]), offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3))
    cum, dh = onp.dm_grid(olk, g["thetas"][0])
    np.testing.assert_allclose(lk.DM_z(g["thetas"][0], zq), onp.interp_hermite(zq, olk.z_grid, cum, dh), rtol=1e-13, atol=1e-10)
    assert lk.log_prior(np.array([-19.3, 70.0, 0.3, 0.0])) == pytest.approx(
        -np.sum(np.log(g["bounds"][:, 1] - g["bounds"][:, 0])) - 0.5 * ((70.0 - 70.39) / 1.80) ** 2)
    lk.engine.close()


# ---- two-component CMB sub-vector (bao/desi_union3_omh2_theta_star.py) ---------------------------------------------------------
def lk_bao_desi_union3_omh2_theta_star(g):
    d = _cmbdata("EARLY_LCDM")
    inv = np.zeros((3, 3))
    inv[np.ix_([0, 2], [0, 2])] = g["inv_cov_cmb_2x2"]
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, offset=onp.Slot(0), H0=onp.Slot(1),
                          obh2=onp.Slot(2), och2=onp.Slot(3), v=onp.Slot(4), z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"],
                          z_turn=0.2, chol=np.linalg.cholesky(g["cov_sn"]), bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True, rd_fit=d["rd_fit"], cmb_mode=3,
                          cmb_prior=d["cmb_prior"], cmb_inv_cov=inv, zstar_fit=d["zstar_fit"], **_phys(d))


def test_oracle_cmb_sub_vector():
    g = golden("bao_desi_union3_omh2_theta_star")
    _check_oracle(lk_bao_desi_union3_omh2_theta_star(g), g)
    # the mirror's embedding of the 2 x 2 inverse = the reference's own matrix
    d = _cmbdata("EARLY_LCDM")
    np.testing.assert_allclose(np.linalg.inv(np.asarray(d["cmb_cov"])[np.ix_([0, 2], [0, 2])]), g["inv_cov_cmb_2x2"], rtol=1e-12)


@pytest.mark.gpu
def test_gpu_cmb_sub_vector(gpu):
    g = golden("bao_desi_union3_omh2_theta_star")
    lk = gpu.likelihoods.DesiUnion3ThetaStarSubset(g["z_cmb"], g["z_hel"], g["obs"], g["cov_sn"], *_bao_args(g), components=(0, 2))
    np.testing.assert_allclose(lk.chi_squared(g["thetas"]), g["chi2"], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"]), g["logl"], rtol=RTOL)
    lk.engine.close()


# ---- three more block combinations: r_drag fit in the late-time flat model, BAO + chronometers, chronometers alone -------------------
def lk_bao_desi_bbn(g):
    d = _cmbdata("PLANCK")
    return onp.Likelihood(ndim=4, z_max=float(g["z_max"]), fde=onp.FDE_THAWING, H0=onp.Slot(0), Om=onp.Slot(1), obh2=onp.Slot(2),
                          w0=onp.Slot(3), bao_z=g["bao_z"], bao_val=g["bao_val"], bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"],
                          rd_fit=d["rd_fit"], rd_wm_late=True, bounds=g["bounds"], gauss=[(2, float(g["bbn"][0]), float(g["bbn"][1]))])


def lk_bao_desi_cc(g):
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), fde=onp.FDE_THAWING, fcc=onp.Slot(0), H0=onp.Slot(1), rd=onp.Slot(2),
                          Om=onp.Slot(3), w0=onp.Slot(4), bao_z=g["bao_z"], bao_val=g["bao_val"], bao_qty=g["bao_qty"],
                          bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True, cc_z=g["cc_z"], cc_h=g["cc_h"],
                          cc_inv_cov=np.linalg.inv(g["cc_cov"]), cc_logdet=np.linalg.slogdet(g["cc_cov"])[1], bounds=g["bounds"])


def lk_ohd_cc(g):
    return onp.Likelihood(ndim=3, z_max=float(np.max(g["cc_z"]) + 0.1), H0=onp.Slot(0), Om=onp.Slot(1), fcc=onp.Slot(2),
                          cc_z=g["cc_z"], cc_h=g["cc_h"], cc_inv_cov=np.linalg.inv(g["cc_cov"]),
                          cc_logdet=np.linalg.slogdet(g["cc_cov"])[1])


def test_planck_compression_constants_match_the_reference_module(pkg):
    g = golden("bao_desi_bbn")
    d = _cmbdata("PLANCK")
    np.testing.assert_allclose([onp.r_drag(d["rd_fit"], 0.0224, 0.143), onp.r_drag(d["rd_fit"], 0.02, 0.12)], g["rdrag_planck"], rtol=1e-14)
    np.testing.assert_allclose([onp.z_star(d["zstar_fit"], 0.0224, 0.143), onp.z_star(d["zstar_fit"], 0.02, 0.12)], g["zstar_planck"], rtol=1e-14)
    np.testing.assert_allclose([d["or_h2"], d["omnu_h2"], d["nu_m0"], d["nu_rho0"]], g["planck_consts"], rtol=1e-14)
    np.testing.assert_array_equal(d["cmb_prior"], g["planck_priors"])
    np.testing.assert_array_equal(d["cmb_cov"], g["planck_cov"])


@pytest.mark.parametrize("name", ["bao_desi_bbn", "bao_desi_cc", "ohd_cc"])
def test_oracle_more_block_combinations(name):
    g = golden(name)
    lk = globals()["lk_" + name](g)
    with np.errstate(all="ignore"):
        for k in range(len(g["thetas"])):
            th = g["thetas"][k]
            if "logp" in g:
                want = g["logp"][k]
                got = onp.log_probability(lk, th)
                assert (got == -np.inf and want == -np.inf) or got == pytest.approx(want, rel=RTOL)
                if not np.isfinite(want):
                    continue
            assert onp.chi_squared(lk, th) == pytest.approx(g["chi2"][k], rel=RTOL)
            if "logl" in g:
                assert onp.log_likelihood(lk, th) == pytest.approx(g["logl"][k], rel=RTOL)


@pytest.mark.gpu
def test_gpu_more_block_combinations(gpu):
    g = golden("bao_desi_bbn")
    lk = gpu.likelihoods.DesiBbn(*_bao_args(g), bounds=g["bounds"], bbn=(float(g["bbn"][0]), float(g["bbn"][1])))
    fin = np.isfinite(g["logp"])
    np.testing.assert_allclose(lk.chi_squared(g["thetas"])[fin], g["chi2"][fin], rtol=RTOL)
    logp = lk.log_probs_vectorized(g["thetas"])
    np.testing.assert_allclose(logp[fin], g["logp"][fin], rtol=RTOL)
    assert np.all(logp[~fin] == -np.inf)
    for k in range(4):
        np.testing.assert_allclose(lk.bao_theory(g["thetas"][k]), g["theory"][k], rtol=1e-12)
    lk.engine.close()

    g = golden("bao_desi_cc")
    lk = gpu.likelihoods.DesiCc(*_bao_args(g), g["cc_z"], g["cc_h"], g["cc_cov"], bounds=g["bounds"])
    fin = np.isfinite(g["logp"])
    np.testing.assert_allclose(lk.chi_squared(g["thetas"])[fin], g["chi2"][fin], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"])[fin], g["logl"][fin], rtol=RTOL)
    logp = lk.log_probs_vectorized(g["thetas"])
    np.testing.assert_allclose(logp[fin], g["logp"][fin], rtol=RTOL)
    assert np.all(logp[~fin] == -np.inf)
    lk.engine.close()

    g = golden("ohd_cc")
    lk = gpu.likelihoods.Cc(g["cc_z"], g["cc_h"], g["cc_cov"])
    np.testing.assert_allclose(lk.chi_squared(g["thetas"]), g["chi2"], rtol=RTOL)
    np.testing.assert_allclose(lk.log_likelihood(g["thetas"]), g["logl"], rtol=RTOL)
    lk.engine.close()
