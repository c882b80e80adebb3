"""Property tests (hypothesis) — SURVEY.md section 4 (4): Hermite reproduces nodes / derivatives, PCHIP == scipy,
forward substitution == Delta^T C^-1 Delta; CPU on the oracle, GPU on the HIP operators."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import oracle_c as oc, oracle_np as onp

SET = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


def _grid(seed, n):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.uniform(0.05, 1.0, n))
    return rng, x


@settings(max_examples=40, **SET)
@given(st.integers(0, 10**6), st.integers(3, 80))
def test_oracle_pchip_equals_scipy_and_is_shape_preserving(seed, n):
    from scipy.interpolate import PchipInterpolator

    rng, x = _grid(seed, n)
    y = np.cumsum(rng.uniform(0.0, 1.0, n))  # monotone data
    xq = rng.uniform(x[0], x[-1], 50)
    got = onp.interp_pchip(xq, x, y)
    np.testing.assert_allclose(got, PchipInterpolator(x, y)(xq), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(oc.interp_pchip(xq, x, y), got, rtol=1e-14)
    order = np.argsort(xq)
    assert np.all(np.diff(got[order]) >= -1e-12), "PCHIP of monotone data is monotone"
    assert np.all(onp.interp_pchip(np.array([x[0] - 1, x[-1] + 1]), x, y) == [y[0], y[-1]])  # clamped outside


@settings(max_examples=40, **SET)
@given(st.integers(0, 10**6), st.integers(2, 60))
def test_oracle_hermite_reproduces_nodes_and_is_exact_for_cubics(seed, n):
    rng, x = _grid(seed, n)
    a, b, c, d = rng.standard_normal(4)
    f = lambda t: a + b * t + c * t**2 + d * t**3
    fp = lambda t: b + 2 * c * t + 3 * d * t**2
    np.testing.assert_allclose(onp.interp_hermite(x[1:-1], x, f(x), fp(x)), f(x[1:-1]), rtol=1e-12, atol=1e-12)
    xq = rng.uniform(x[0], x[-1], 30)
    np.testing.assert_allclose(onp.interp_hermite(xq, x, f(x), fp(x)), f(xq), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(oc.interp_hermite(xq, x, f(x), fp(x)), onp.interp_hermite(xq, x, f(x), fp(x)), rtol=1e-14, atol=1e-14)


@settings(max_examples=25, **SET)
@given(st.integers(0, 10**6), st.integers(1, 120))
def test_oracle_forward_substitution_is_the_quadratic_form(seed, n):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    C = M @ M.T + n * np.eye(n)
    L = np.linalg.cholesky(C) + np.triu(rng.standard_normal((n, n)), 1)  # junk above the diagonal is never read
    b = rng.standard_normal(n)
    ref = b @ np.linalg.solve(C, b)
    assert onp.solve_triangular_chi2(L, b) == pytest.approx(ref, rel=1e-9)
    assert oc.solve_triangular(L, b) == pytest.approx(ref, rel=1e-9)


@settings(max_examples=25, **SET)
@given(st.integers(0, 10**6), st.integers(1, 400), st.integers(1, 40))
def test_host_packings_agree_with_forward_substitution(pkg, seed, n, dummy):
    """Both packed layouts (blocked streams, explicit-inverse streams) replayed on the host for random sizes."""
    import ctypes as C

    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    L = np.linalg.cholesky(M @ M.T + n * np.eye(n))
    b = rng.standard_normal(n)
    ref = oc.solve_triangular(L, b)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    chi2 = C.c_double()
    pkg._lib.check(pkg.lib().cf_selftest_pack_host(p(L), n, n, p(b), C.byref(chi2), None))
    assert chi2.value == pytest.approx(ref, rel=1e-11)
    pkg._lib.check(pkg.lib().cf_selftest_invpack_host(p(L), n, n, p(b), C.byref(chi2), None))
    assert chi2.value == pytest.approx(ref, rel=1e-11)


@pytest.mark.gpu
@settings(max_examples=12, **SET)
@given(st.integers(0, 10**6), st.integers(1, 700), st.integers(1, 70))
def test_gpu_solve_triangular_random_shapes(pkg, seed, n, nrhs):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("needs an MI355X")
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    L = np.linalg.cholesky(M @ M.T + n * np.eye(n)) + np.triu(rng.standard_normal((n, n)), 1) * 2.0
    b = rng.standard_normal((nrhs, n))
    got = pkg.solve_triangular.solve_triangular(L, b)
    ref = np.array([oc.solve_triangular(L, bb) for bb in b])
    np.testing.assert_allclose(got, ref, rtol=1e-11)


@pytest.mark.gpu
@settings(max_examples=12, **SET)
@given(st.integers(0, 10**6), st.integers(3, 300), st.integers(1, 200))
def test_gpu_interpolators_random_grids(pkg, seed, n, nq):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("needs an MI355X")
    rng, x = _grid(seed, n)
    y, yp = rng.standard_normal(n), rng.standard_normal(n)
    xq = np.concatenate([rng.uniform(x[0] - 1, x[-1] + 1, nq), x[:: max(1, n // 7)]])
    np.testing.assert_allclose(pkg.interpolator.interp_hermite(xq, x, y, yp), onp.interp_hermite(xq, x, y, yp), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(pkg.interpolator.interp_pchip(xq, x, y), onp.interp_pchip(xq, x, y), rtol=1e-12, atol=1e-13)


@pytest.mark.gpu
@settings(max_examples=8, **SET)
@given(st.integers(0, 10**6), st.integers(1, 400), st.integers(1, 90), st.sampled_from(["inverse", "blocked"]))
def test_gpu_sn_likelihood_random_sizes_both_solves(pkg, seed, n_sn, n_walkers, solve):
    """Full path (walker kernel + either solve kernel) at ragged sizes: N not a multiple of 16 / 64, W not a multiple
    of 16 / 32, against the C oracle at the 1e-10 bar."""
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("needs an MI355X")
    from oracle import oracle_np as onp

    syn = pkg.synthetic.pantheon_like(n_sn=n_sn, seed=seed)
    lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"], solve=solve)
    ref = oc.COracle(onp.Likelihood(
        ndim=4, z_max=lk.z_max, offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3),
        z_cmb=syn["z_cmb"], z_hel=syn["z_hel"], obs=syn["obs"], chol=syn["chol"],
        bounds=pkg.sn_pantheon.bounds, gauss=[pkg.sn_pantheon.H0_PRIOR]))
    theta = pkg.synthetic.walkers(pkg.sn_pantheon.bounds, n_walkers, seed=seed + 1)
    np.testing.assert_allclose(lk.chi_squared(theta), ref.chi2(theta), rtol=1e-10)
    np.testing.assert_allclose(lk.log_probs_vectorized(theta), ref.logp(theta), rtol=1e-10)
    lk.engine.close()
