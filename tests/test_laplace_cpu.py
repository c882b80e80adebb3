"""CPU: batched Laplace evidence (the job of the reference's log_evidence.py) on analytic targets."""
import numpy as np
import pytest


def test_hessian_and_gradient_on_a_quadratic(pkg):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((4, 4))
    P = A @ A.T + 4 * np.eye(4)  # precision matrix
    mu = np.array([0.3, -1.0, 2.0, 0.5])
    calls = []

    def logp(batch):
        calls.append(len(batch))
        d = batch - mu
        return -0.5 * np.einsum("wi,ij,wj->w", d, P, d)

    h = np.full(4, 1e-3)
    val, g = pkg.laplace.gradient(logp, mu + 0.1, h)
    np.testing.assert_allclose(g, -P @ np.full(4, 0.1), rtol=1e-8)
    H = pkg.laplace.hessian(logp, mu + 0.05, h)
    np.testing.assert_allclose(H, -P, rtol=1e-6, atol=1e-6)
    assert calls == [9, 65], "one batch of 2n+1 points, one batch of 4n^2+1 points"


def test_log_evidence_of_a_normalised_gaussian_is_zero(pkg):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((3, 3))
    cov = A @ A.T + np.eye(3)
    P = np.linalg.inv(cov)
    mu = np.array([1.0, -2.0, 0.5])
    norm = -0.5 * (3 * np.log(2 * np.pi) + np.linalg.slogdet(cov)[1])
    bounds = np.array([(-20.0, 20.0)] * 3)

    def logp(batch):
        inside = np.all((batch > bounds[:, 0]) & (batch < bounds[:, 1]), axis=1)
        d = batch - mu
        return np.where(inside, norm - 0.5 * np.einsum("wi,ij,wj->w", d, P, d), -np.inf)

    samples = rng.multivariate_normal(mu, cov, size=500)
    lnz, det = pkg.laplace.log_evidence(samples, logp(samples), logp, bounds, return_details=True)
    assert lnz == pytest.approx(0.0, abs=1e-5)  # the Laplace approximation is exact for a Gaussian
    np.testing.assert_allclose(det["theta_map"], mu, atol=1e-5)
    np.testing.assert_allclose(det["hessian"], -P, rtol=1e-5, atol=1e-7)


def test_log_evidence_near_a_box_edge_keeps_the_stencil_inside(pkg):
    bounds = np.array([(0.0, 1.0), (-1.0, 1.0)])

    def logp(batch):
        inside = np.all((batch > bounds[:, 0]) & (batch < bounds[:, 1]), axis=1)
        return np.where(inside, -0.5 * (((batch - np.array([0.001, 0.0])) / 0.05) ** 2).sum(axis=1), -np.inf)

    s = np.array([[0.002, 0.01], [0.01, -0.02]])
    lnz = pkg.laplace.log_evidence(s, logp(s), logp, bounds)
    assert np.isfinite(lnz) and lnz == pytest.approx(np.log(2 * np.pi * 0.05**2), abs=1e-3)
