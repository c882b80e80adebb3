"""CPU: the numpy oracle (oracle/oracle_np.py) against golden vectors produced by the reference itself."""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle_np as onp


def test_linspace_grid_is_bit_exact():
    # the kernels rebuild z_grid as i*step with the last node forced to z_max (numpy.linspace)
    for zmax, G in ((2.36137, 4000), (2.43, 4000), (1.0, 7), (4.26, 4000)):
        ref = np.linspace(0, zmax, num=G)
        step = zmax / (G - 1)
        mine = np.arange(G) * step
        mine[-1] = zmax
        assert np.array_equal(ref, mine)


def test_interp_hermite_golden():
    g = golden("interpolator")
    out = onp.interp_hermite(g["h_xq"], g["h_x"], g["h_y"], g["h_yp"])
    np.testing.assert_allclose(out, g["h_out"], rtol=1e-15, atol=0)


def test_pchip_golden():
    g = golden("interpolator")
    np.testing.assert_allclose(onp.pchip_slopes(g["p_x"], g["p_y"]), g["p_slopes"], rtol=1e-15, atol=1e-300)
    np.testing.assert_allclose(onp.interp_pchip(g["p_xq"], g["p_x"], g["p_y"]), g["p_out"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(onp.interp_pchip(g["m_xq"], g["m_x"], g["m_y"]), g["m_out"], rtol=1e-15)
    for name in ("p3", "p4"):
        np.testing.assert_allclose(onp.pchip_slopes(g[name + "_x"], g[name + "_y"]), g[name + "_slopes"], rtol=1e-15)


def test_pchip_matches_scipy():
    from scipy.interpolate import PchipInterpolator

    g = golden("interpolator")
    xq = g["m_xq"][(g["m_xq"] > g["m_x"][0]) & (g["m_xq"] < g["m_x"][-1])]
    np.testing.assert_allclose(onp.interp_pchip(xq, g["m_x"], g["m_y"]), PchipInterpolator(g["m_x"], g["m_y"])(xq), rtol=1e-14)


def test_solve_triangular_golden():
    g = golden("interpolator")
    out = np.array([onp.solve_triangular_chi2(g["t_L"], b) for b in g["t_b"]])
    np.testing.assert_allclose(out, g["t_out"], rtol=1e-14)
    # and it really is b^T C^-1 b, ignoring the garbage above the diagonal
    Lc = np.tril(g["t_L"])
    C = Lc @ Lc.T
    np.testing.assert_allclose(out, [b @ np.linalg.solve(C, b) for b in g["t_b"]], rtol=1e-11)


def _pantheon_lk(g):
    return onp.Likelihood(
        ndim=4, z_max=float(g["z_max"]), offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3),
        z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], z_turn=0.15, chol=g["chol"], bounds=g["bounds"],
        gauss=[(1, 70.39, 1.80)],
    )


def test_sn_pantheon_intermediates(pantheon_golden):
    g = pantheon_golden
    lk = _pantheon_lk(g)
    np.testing.assert_array_equal(lk.z_grid[::250], g["z_grid_sub"])
    np.testing.assert_array_equal(lk.dz[::250], g["dz_sub"])
    for k in range(3):
        th = g["thetas"][k]
        cum, dh = onp.dm_grid(lk, th)
        np.testing.assert_allclose(dh[::250], g[f"dh_sub_{k}"], rtol=4e-16)
        np.testing.assert_allclose(cum[::250], g[f"cum_sub_{k}"], rtol=1e-15)
        DM, mucorr, muth, delta = onp.sn_parts(lk, th)
        np.testing.assert_allclose(DM, g[f"dm_{k}"], rtol=1e-15)
        np.testing.assert_allclose(muth, g[f"muth_{k}"], rtol=1e-15)
        np.testing.assert_allclose(mucorr, g[f"mucorr_{k}"], rtol=0, atol=2e-15)
        np.testing.assert_allclose(delta, g[f"delta_{k}"], rtol=0, atol=3e-14)


def test_sn_pantheon_chi2_logp(pantheon_golden):
    g = pantheon_golden
    lk = _pantheon_lk(g)
    finite = np.isfinite(g["logp"])
    assert finite.sum() >= 25 and (~finite).sum() >= 8
    for th, chi2, logp in zip(g["thetas"], g["chi2"], g["logp"]):
        lp = onp.log_probability(lk, th)
        if np.isfinite(logp):
            assert onp.chi_squared(lk, th) == pytest.approx(chi2, rel=1e-12)
            assert lp == pytest.approx(logp, rel=1e-12)
        else:
            assert lp == -np.inf


# ---- the C restatement (oracle/cosmofit_oracle.c) against the same golden vectors ----------
def test_c_oracle_interpolator_and_trsv():
    from oracle import oracle_c as oc

    g = golden("interpolator")
    np.testing.assert_allclose(oc.interp_hermite(g["h_xq"], g["h_x"], g["h_y"], g["h_yp"]), g["h_out"], rtol=1e-15)
    np.testing.assert_allclose(oc.pchip_slopes(g["p_x"], g["p_y"]), g["p_slopes"], rtol=1e-15, atol=1e-300)
    np.testing.assert_allclose(oc.interp_pchip(g["p_xq"], g["p_x"], g["p_y"]), g["p_out"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(oc.interp_pchip(g["m_xq"], g["m_x"], g["m_y"]), g["m_out"], rtol=1e-15)
    for name in ("p3", "p4"):
        np.testing.assert_allclose(oc.pchip_slopes(g[name + "_x"], g[name + "_y"]), g[name + "_slopes"], rtol=1e-15)
    out = [oc.solve_triangular(g["t_L"], b) for b in g["t_b"]]
    np.testing.assert_allclose(out, g["t_out"], rtol=1e-14)


def test_c_oracle_sn_pantheon(pantheon_golden):
    from oracle import oracle_c as oc

    g = pantheon_golden
    co = oc.COracle(_pantheon_lk(g))
    for k in range(3):
        p = co.sn_parts(g["thetas"][k])
        np.testing.assert_allclose(p["dh"][::250], g[f"dh_sub_{k}"], rtol=4e-16)
        np.testing.assert_allclose(p["cum"][::250], g[f"cum_sub_{k}"], rtol=1e-15)
        np.testing.assert_allclose(p["dm"], g[f"dm_{k}"], rtol=1e-15)
        np.testing.assert_allclose(p["mu_corr"], g[f"mucorr_{k}"], rtol=0, atol=2e-15)
        np.testing.assert_allclose(p["delta"], g[f"delta_{k}"], rtol=0, atol=3e-14)
    finite = np.isfinite(g["logp"])
    chi2 = co.chi2(g["thetas"][finite])
    np.testing.assert_allclose(chi2, g["chi2"][finite], rtol=1e-12)
    logp = co.logp(g["thetas"])
    np.testing.assert_allclose(logp[finite], g["logp"][finite], rtol=1e-12)
    assert np.all(logp[~finite] == -np.inf)
    # thread count must not change any value (walkers are independent)
    np.testing.assert_array_equal(co.chi2(g["thetas"][finite], nthreads=1), chi2)


# ---- BAO / compressed-CMB / physical-density blocks ------------------------------------------------------
from conftest import load_pkg


def _cmbdata(name):
    return getattr(load_pkg().cmb_data, name)


@pytest.mark.parametrize("fixture,comp", [("bao_desi_cmb", "EARLY_LCDM"), ("bao_desi_fs_lya_cmb", "PLANCK_ACT")])
def test_cmb_constants_match_the_reference_modules(fixture, comp):
    g, d = golden(fixture), _cmbdata(comp)
    for key in ("or_h2", "omnu_h2", "o_gamma_h2", "nu_m0", "nu_rho0"):
        assert d[key] == pytest.approx(float(g[key]), rel=1e-15), key
    np.testing.assert_allclose(d["nu_qs_sq"], g["nu_qs_sq"], rtol=1e-15)
    np.testing.assert_allclose(d["nu_ws"], g["nu_ws"], rtol=0)
    np.testing.assert_allclose(d["cmb_prior"], g["cmb_priors"], rtol=0)
    np.testing.assert_allclose(d["cmb_inv_cov"], g["cmb_inv_cov"], rtol=1e-13)
    lk = onp.Likelihood(ndim=1, z_max=1.0, **{k: d[k] for k in ("nu_m0", "nu_rho0", "nu_qs_sq", "nu_ws")})
    np.testing.assert_allclose(onp.Omnu_z(lk, g["omnu_z_probe"]), g["omnu_z_vals"], rtol=1e-15)
    for wb, wm, zs, rdv in zip(g["fit_wb"], g["fit_wm"], g["zstar_vals"], g["rdrag_vals"]):
        assert onp.z_star(d["zstar_fit"], wb, wm) == pytest.approx(zs, rel=1e-15)
        assert onp.r_drag(d["rd_fit"], wb, wm) == pytest.approx(rdv, rel=1e-14)
    gx, gw = np.polynomial.legendre.leggauss(100)
    np.testing.assert_array_equal(gx, g["gl_x"])
    np.testing.assert_array_equal(gw, g["gl_w"])


def _phys(d):
    return {k: d[k] for k in ("or_h2", "omnu_h2", "o_gamma_h2", "nu_m0", "nu_rho0", "nu_qs_sq", "nu_ws")}


def lk_bao_desi(g):
    return onp.Likelihood(ndim=3, z_max=float(g["z_max"]), fde=onp.FDE_THAWING, H0=onp.Slot(0, 100.0), Om=onp.Slot(1),
                          w0=onp.Slot(2), rd=onp.Slot(fixed=float(g["rd"])), bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], bounds=g["bounds"])


def lk_bao_desi_cmb(g):
    d = _cmbdata("EARLY_LCDM")
    return onp.Likelihood(ndim=4, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_THAWING, H0=onp.Slot(0),
                          obh2=onp.Slot(1), och2=onp.Slot(2), w0=onp.Slot(3), bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True, rd_fit=d["rd_fit"],
                          cmb_mode=3, cmb_prior=d["cmb_prior"], cmb_inv_cov=d["cmb_inv_cov"], zstar_fit=d["zstar_fit"],
                          bounds=g["bounds"], **_phys(d))


def lk_bao_desi_fs_lya_cmb(g):
    d = _cmbdata("PLANCK_ACT")
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_CPL, H0=onp.Slot(0),
                          obh2=onp.Slot(1), och2=onp.Slot(2), w0=onp.Slot(3), wa=onp.Slot(4), bao_z=g["bao_z"],
                          bao_val=g["bao_val"], bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], rd_fit=d["rd_fit"],
                          cmb_mode=1, cmb_prior=d["cmb_prior"], cmb_inv_cov=d["cmb_inv_cov"], zstar_fit=d["zstar_fit"],
                          cpl_wall=True, **_phys(d))


def lk_bao_desi_cmb_des5y(g, chol):
    d = _cmbdata("PLANCK_ACT")
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_LCDM, offset=onp.Slot(0),
                          H0=onp.Slot(1), obh2=onp.Slot(2), och2=onp.Slot(3), v=onp.Slot(4), z_cmb=g["z_cmb"],
                          z_hel=g["z_hel"], obs=g["obs"], z_turn=0.10563, chol=chol, bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], rd_fit=d["rd_fit"], cmb_mode=1,
                          cmb_prior=d["cmb_prior"], cmb_inv_cov=d["cmb_inv_cov"], zstar_fit=d["zstar_fit"], **_phys(d))


def lk_bao_desi_cmb_des5y_H0trgb(g, chol):
    d = _cmbdata("PLANCK_ACT")
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_LCDM, offset=onp.Slot(0),
                          H0=onp.Slot(1), obh2=onp.Slot(2), och2=onp.Slot(3), v=onp.Slot(4), z_cmb=g["z_cmb"],
                          z_hel=g["z_hel"], obs=g["obs"], z_turn=0.11, chol=chol, bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True, rd_fit=d["rd_fit"], cmb_mode=1,
                          cmb_prior=d["cmb_prior"], cmb_inv_cov=d["cmb_inv_cov"], zstar_fit=d["zstar_fit"],
                          chi2_gauss=[(1, float(g["h0_prior"][0]), float(g["h0_prior"][1]))], **_phys(d))


def lk_bao_desi_cmb_pantheon(g, chol):
    d = _cmbdata("PLANCK_ACT")
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_LCDM, offset=onp.Slot(0),
                          H0=onp.Slot(1), obh2=onp.Slot(2), och2=onp.Slot(3), v=onp.Slot(4), z_cmb=g["z_cmb"],
                          z_hel=g["z_hel"], obs=g["obs"], z_turn=0.15, chol=chol, bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True, rd_fit=d["rd_fit"], cmb_mode=1,
                          cmb_prior=d["cmb_prior"], cmb_inv_cov=d["cmb_inv_cov"], zstar_fit=d["zstar_fit"], **_phys(d))


def lk_bao_desi_des5y_bbn_theta_star(g, chol):
    d = _cmbdata("PLANCK_ACT")
    inv = np.zeros((3, 3))
    inv[1, 1] = 1.0 / d["cmb_cov"][1, 1]
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_THAWING, offset=onp.Slot(0),
                          H0=onp.Slot(1), obh2=onp.Slot(2), och2=onp.Slot(3), w0=onp.Slot(4), z_cmb=g["z_cmb"],
                          z_hel=g["z_hel"], obs=g["obs"], has_vstep=False, chol=chol, bao_z=g["bao_z"], bao_val=g["bao_val"],
                          bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], bao_dh_exact=True, rd_fit=d["rd_fit"],
                          cmb_mode=2, cmb_prior=d["cmb_prior"], cmb_inv_cov=inv, zstar_fit=d["zstar_fit"],
                          bounds=g["bounds"], gauss=[(2, float(g["bbn"][0]), float(g["bbn"][1]))], **_phys(d))


def _chol_of(g):
    from scipy.linalg import cho_factor
    from conftest import synthetic_cov
    return cho_factor(synthetic_cov(g["sigma"]), lower=True)[0]


def test_oracle_bao_desi():
    g = golden("bao_desi")
    lk = lk_bao_desi(g)
    for k in range(4):
        np.testing.assert_allclose(onp.bao_theory(lk, g["thetas"][k]), g["theory"][k], rtol=1e-13)
    fin = np.isfinite(g["logp"])
    for th, c2, lp in zip(g["thetas"], g["chi2"], g["logp"]):
        got = onp.log_probability(lk, th)
        if np.isfinite(lp):
            assert onp.chi_squared(lk, th) == pytest.approx(c2, rel=1e-11)
            assert got == pytest.approx(lp, rel=1e-11)
        else:
            assert got == -np.inf
    assert fin.sum() >= 18
    # the reference's batch wrapper returns float32 (bao/desi.py:103): ours is the float64 superset
    np.testing.assert_allclose(onp.log_probs_vectorized(lk, g["thetas"])[fin].astype(np.float32), g["logp_vec32"][fin], rtol=1e-6)


def test_oracle_bao_desi_cmb():
    g = golden("bao_desi_cmb")
    lk = lk_bao_desi_cmb(g)
    for k in range(4):
        th = g["thetas"][k]
        np.testing.assert_allclose(onp.H_z(lk, np.array([0.0, 0.7, 2.33, 1089.0, 1.0e5]), th), g["hz_probe"][k], rtol=1e-14)
        np.testing.assert_allclose(onp.cmb_distances(lk, th), g["cmb_dist"][k], rtol=1e-13)
        np.testing.assert_allclose(onp.bao_theory(lk, th), g["theory"][k], rtol=1e-13)
    for th, c2, lp in zip(g["thetas"], g["chi2"], g["logp"]):
        if np.isfinite(lp):
            assert onp.chi_squared(lk, th) == pytest.approx(c2, rel=1e-10)
            assert onp.log_probability(lk, th) == pytest.approx(lp, rel=1e-10)
        else:
            assert onp.log_probability(lk, th) == -np.inf


def test_oracle_bao_desi_fs_lya_cmb():
    g = golden("bao_desi_fs_lya_cmb")
    lk = lk_bao_desi_fs_lya_cmb(g)
    for k in range(4):
        th = g["thetas"][k]
        np.testing.assert_allclose(onp.cmb_distances(lk, th), g["cmb_dist"][k], rtol=1e-13)
        np.testing.assert_allclose(onp.bao_theory(lk, th), g["theory"][k], rtol=1e-12)
    for k in range(8):
        th = g["thetas"][k]
        assert onp.chi2_cmb(lk, th) == pytest.approx(g["chi2_cmb"][k], rel=1e-9)
        assert onp.chi2_bao(lk, th) == pytest.approx(g["chi2_bao"][k], rel=1e-10)
    with np.errstate(all="ignore"):
        for th, ll in zip(g["thetas"], g["logl"]):
            got = onp.log_likelihood(lk, th)
            if ll == -1e8:
                assert got == -1e8
            elif np.isfinite(ll):
                assert got == pytest.approx(ll, rel=1e-9)
    assert (g["logl"] == -1e8).sum() >= 3


def test_oracle_bao_desi_cmb_des5y():
    g = golden("bao_desi_cmb_des5y")
    lk = lk_bao_desi_cmb_des5y(g, _chol_of(g))
    for k in range(4):
        th = g["thetas"][k]
        np.testing.assert_allclose(onp.cmb_distances(lk, th), g["cmb_dist"][k], rtol=1e-13)
        np.testing.assert_allclose(onp.bao_theory(lk, th), g["theory"][k], rtol=1e-12)
        np.testing.assert_allclose(onp.chi2_blocks(lk, th), g["chi2_parts"][k], rtol=1e-10)
    for th, c2, ll in zip(g["thetas"][:8], g["chi2"][:8], g["logl"][:8]):
        assert onp.chi_squared(lk, th) == pytest.approx(c2, rel=1e-11)
        assert onp.log_likelihood(lk, th) == pytest.approx(ll, rel=1e-11)


def test_oracle_bao_desi_cmb_des5y_h0trgb():
    """bao/desi_cmb_des5y_H0trgb.py: two BAO blocks (DESI + the 6dF point) as one block-diagonal block + TRGB H0 term."""
    g = golden("bao_desi_cmb_des5y_H0trgb")
    lk = lk_bao_desi_cmb_des5y_H0trgb(g, _chol_of(g))
    for k in range(4):
        th = g["thetas"][k]
        np.testing.assert_allclose(onp.cmb_distances(lk, th), g["cmb_dist"][k], rtol=1e-13)
        np.testing.assert_allclose(onp.bao_theory(lk, th), g["theory"][k], rtol=1e-12)
    for th, c2, ll in zip(g["thetas"][:8], g["chi2"][:8], g["logl"][:8]):
        assert onp.chi_squared(lk, th) == pytest.approx(c2, rel=1e-11)
        assert onp.log_likelihood(lk, th) == pytest.approx(ll, rel=1e-11)


def lk_sn_cmb(g, chol, z_turn, bounds=None):
    d = _cmbdata("PLANCK_ACT")
    return onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_LCDM, offset=onp.Slot(0),
                          H0=onp.Slot(1), obh2=onp.Slot(2), och2=onp.Slot(3), v=onp.Slot(4), z_cmb=g["z_cmb"],
                          z_hel=g["z_hel"], obs=g["obs"], z_turn=z_turn, chol=chol, cmb_mode=1, cmb_prior=d["cmb_prior"],
                          cmb_inv_cov=d["cmb_inv_cov"], zstar_fit=d["zstar_fit"], bounds=bounds, **_phys(d))


def test_oracles_sn_des5y_and_the_two_sn_cmb_scripts():
    """sn/des5y.py (flat LCDM on DES-Dovekie, step at z = 0.11), sn/des5y_cmb.py and sn/pantheon_cmb.py (SN + Planck/ACT)."""
    from oracle import oracle_c as oc

    g = golden("sn_des5y")
    lk = onp.Likelihood(ndim=4, z_max=float(g["z_max"]), offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3),
                        z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], z_turn=0.11, chol=_chol_of(g))
    for th, c2 in zip(g["thetas"][:4], g["chi2"][:4]):
        assert onp.chi_squared(lk, th) == pytest.approx(c2, rel=1e-11)
    co = oc.COracle(lk)
    np.testing.assert_allclose(co.chi2(g["thetas"]), g["chi2"], rtol=1e-10)
    np.testing.assert_allclose(co.logl(g["thetas"]), g["logl"], rtol=1e-10)

    g = golden("sn_des5y_cmb")
    lk = lk_sn_cmb(g, _chol_of(g), 0.11)
    for k in range(3):
        np.testing.assert_allclose(onp.cmb_distances(lk, g["thetas"][k]), g["cmb_dist"][k], rtol=1e-13)
        assert onp.chi_squared(lk, g["thetas"][k]) == pytest.approx(g["chi2"][k], rel=1e-11)
    co = oc.COracle(lk)
    np.testing.assert_allclose(co.chi2(g["thetas"]), g["chi2"], rtol=1e-10)

    g = golden("sn_pantheon_cmb")
    lk = lk_sn_cmb(g, _chol_of(g), 0.15, bounds=g["bounds"])
    fin = np.isfinite(g["logp"])
    co = oc.COracle(lk)
    got = co.logp(g["thetas"])
    np.testing.assert_allclose(got[fin], g["logp"][fin], rtol=1e-10)
    assert np.all(got[~fin] == -np.inf) and (~fin).sum() >= 10
    k = int(np.flatnonzero(fin)[0])
    assert onp.log_probability(lk, g["thetas"][k]) == pytest.approx(g["logp"][k], rel=1e-11)


def test_oracles_ohd_cc_des5y():
    """ohd/cc_des5y.py: late-time flat wCDM, SN without velocity step, chronometers with f_cc and the log-det term."""
    from oracle import oracle_c as oc

    g = golden("ohd_cc_des5y")
    lk = onp.Likelihood(ndim=5, z_max=float(g["z_max"]), ez_model=onp.EZ_LATE_FLAT, fde=onp.FDE_WCDM, fcc=onp.Slot(0),
                        offset=onp.Slot(1), H0=onp.Slot(2), Om=onp.Slot(3), w0=onp.Slot(4), z_cmb=g["z_cmb"], z_hel=g["z_hel"],
                        obs=g["obs"], has_vstep=False, chol=_chol_of(g), cc_z=g["cc_z"], cc_h=g["cc_h"],
                        cc_inv_cov=g["cc_inv_cov"], cc_logdet=float(g["cc_logdet"]), bounds=g["bounds"])
    fin = np.isfinite(g["logp"])
    for k in np.flatnonzero(fin)[:4]:
        assert onp.chi_squared(lk, g["thetas"][k]) == pytest.approx(g["chi2"][k], rel=1e-11)
        assert onp.log_likelihood(lk, g["thetas"][k]) == pytest.approx(g["logl"][k], rel=1e-11)
        assert onp.log_probability(lk, g["thetas"][k]) == pytest.approx(g["logp"][k], rel=1e-11)
    co = oc.COracle(lk)
    got = co.logp(g["thetas"])
    np.testing.assert_allclose(got[fin], g["logp"][fin], rtol=1e-10)
    assert np.all(got[~fin] == -np.inf) and (~fin).sum() >= 10
    np.testing.assert_allclose(co.chi2(g["thetas"][fin]), g["chi2"][fin], rtol=1e-10)


def test_oracle_bao_desi_des5y_bbn_theta_star():
    g = golden("bao_desi_des5y_bbn_theta_star")
    lk = lk_bao_desi_des5y_bbn_theta_star(g, _chol_of(g))
    for k in range(4):
        np.testing.assert_allclose(onp.bao_theory(lk, g["thetas"][k]), g["theory"][k], rtol=1e-12)
    n = 0
    for th, c2, lp in zip(g["thetas"], g["chi2"], g["logp"]):
        if np.isfinite(lp):
            if n < 6:
                assert onp.chi_squared(lk, th) == pytest.approx(c2, rel=1e-11)
                assert onp.log_probability(lk, th) == pytest.approx(lp, rel=1e-11)
                n += 1
        else:
            assert onp.log_probability(lk, th) == -np.inf


# ---- the C restatement on the joint likelihoods -----------------------------------------------------------
@pytest.mark.parametrize("name", ["bao_desi", "bao_desi_cmb", "bao_desi_fs_lya_cmb", "bao_desi_cmb_des5y",
                                  "bao_desi_cmb_des5y_H0trgb", "bao_desi_cmb_pantheon", "bao_desi_des5y_bbn_theta_star"])
def test_c_oracle_joint_likelihoods(name):
    from oracle import oracle_c as oc

    g = golden(name)
    build = globals()["lk_" + name]
    lk = build(g, _chol_of(g)) if "z_cmb" in g else build(g)
    co = oc.COracle(lk)
    for k in range(4):
        blocks, theory, cmbv = co.blocks(g["thetas"][k])
        np.testing.assert_allclose(theory, g["theory"][k], rtol=1e-12)
        if "cmb_dist" in g:
            np.testing.assert_allclose(cmbv, g["cmb_dist"][k], rtol=1e-13)
        if "chi2_parts" in g:
            np.testing.assert_allclose(blocks, g["chi2_parts"][k], rtol=1e-10)
    with np.errstate(all="ignore"):
        if "logp" in g:
            fin = np.isfinite(g["logp"])
            got = co.logp(g["thetas"])
            np.testing.assert_allclose(got[fin], g["logp"][fin], rtol=1e-10)
            assert np.all(got[~fin] == -np.inf)
            np.testing.assert_allclose(co.chi2(g["thetas"][fin]), g["chi2"][fin], rtol=1e-10)
        if "logl" in g:
            fin = np.isfinite(g["logl"])
            got = co.logl(g["thetas"])
            np.testing.assert_allclose(got[fin], g["logl"][fin], rtol=1e-9)


# ---- Union3 (explicit inverse -> Cholesky), dipole weights, SH0ES calibrators, cosmic chronometers
def lk_sn_union3_1(g):
    return onp.Likelihood(ndim=3, z_max=float(g["z_max"]), offset=onp.Slot(0), H0=onp.Slot(fixed=float(g["H0"])), Om=onp.Slot(1),
                          v=onp.Slot(2), z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], z_turn=0.2,
                          chol=np.linalg.cholesky(g["cov"]))


def lk_sn_pantheon_dipole(g):
    return onp.Likelihood(ndim=4, z_max=float(g["z_max"]), offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3),
                          z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], step=g["weights"], chol=_chol_of(g))


def lk_sn_pantheon_and_sh0es(g):
    fixed = np.where(g["ceph"] != -9, g["ceph"], np.nan)
    return onp.Likelihood(ndim=4, z_max=float(g["z_max"]), offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3),
                          z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], step=g["corr_sign"], fixed_mu=fixed,
                          chol=_chol_of(g), bounds=g["bounds"])


def lk_bao_desi_union3_cc_theta_star(g):
    d = _cmbdata("PLANCK_ACT")
    inv = np.zeros((3, 3))
    inv[1, 1] = 1.0 / d["cmb_cov"][1, 1]
    return onp.Likelihood(ndim=6, z_max=float(g["z_max"]), ez_model=onp.EZ_PHYSICAL, fde=onp.FDE_LCDM, fcc=onp.Slot(0),
                          offset=onp.Slot(1), H0=onp.Slot(2), obh2=onp.Slot(3), och2=onp.Slot(4), v=onp.Slot(5),
                          z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], z_turn=0.2, chol=np.linalg.cholesky(g["cov_sn"]),
                          bao_z=g["bao_z"], bao_val=g["bao_val"], bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"],
                          bao_dh_exact=True, rd_fit=d["rd_fit"], cmb_mode=2, cmb_prior=d["cmb_prior"], cmb_inv_cov=inv,
                          zstar_fit=d["zstar_fit"], cc_z=g["cc_z"], cc_h=g["cc_h"], cc_inv_cov=g["cc_inv_cov"],
                          cc_logdet=float(g["cc_logdet"]), **_phys(d))


@pytest.mark.parametrize("name", ["sn_union3_1", "sn_pantheon_dipole", "sn_pantheon_and_sh0es", "bao_desi_union3_cc_theta_star"])
def test_oracles_on_widening_scripts(name):
    from oracle import oracle_c as oc

    g = golden(name)
    lk = globals()["lk_" + name](g)
    co = oc.COracle(lk)
    with np.errstate(all="ignore"):
        if "logp" in g:
            fin = np.isfinite(g["logp"])
            got = co.logp(g["thetas"])
            np.testing.assert_allclose(got[fin], g["logp"][fin], rtol=1e-10)
            assert np.all(got[~fin] == -np.inf)
        fin = np.isfinite(g["chi2"])
        np.testing.assert_allclose(co.chi2(g["thetas"])[fin], g["chi2"][fin], rtol=1e-10)
        if "logl" in g:
            np.testing.assert_allclose(co.logl(g["thetas"])[fin], g["logl"][fin], rtol=1e-10)
        k = int(np.flatnonzero(fin)[0])
        assert onp.chi_squared(lk, g["thetas"][k]) == pytest.approx(g["chi2"][k], rel=1e-10)
        if "logl" in g:
            assert onp.log_likelihood(lk, g["thetas"][k]) == pytest.approx(g["logl"][k], rel=1e-10)
    if name == "sn_union3_1":  # the reference's docstring chi^2 at its posterior medians: 28.76 (sn/union3_1.py:145)
        assert g["chi2"][-1] == pytest.approx(28.76, abs=0.01)
