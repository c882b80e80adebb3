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
