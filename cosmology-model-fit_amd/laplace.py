"""
Batched Laplace evidence: the job of the reference's ``log_evidence.py`` (L-BFGS-B polish of the best sample, then
ln Z = log P(MAP) + (n/2) ln 2 pi - (1/2) ln det(-H), log_evidence.py:7-70) with every finite-difference stencil
evaluated as ONE batch of the vectorised log-probability instead of hundreds of serial single-theta calls
(SURVEY.md 8f-3).  Same call signature; ``log_probability`` must accept a batch [W, ndim] -> [W].

* MAP polish: scipy's L-BFGS-B as in the reference (log_evidence.py:26), but value + central-difference gradient
  come from one batched call of 2 n + 1 points per iteration.
* Hessian: central second differences at steps h and 2h combined by Richardson extrapolation ((4 H_h - H_2h) / 3),
  4 n^2 + 1 points in a single batch (the reference uses numdifftools.Hessian(step=1e-5), also a Richardson scheme,
  evaluated one point at a time, log_evidence.py:45-46).
* jitter / slogdet / NaN conventions: log_evidence.py:49-66.
"""
import numpy as np
from scipy.optimize import minimize


def _batched(log_probability, pts):
    out = np.asarray(log_probability(np.ascontiguousarray(pts, dtype=np.float64)), dtype=np.float64)
    if out.shape != (len(pts),):
        raise ValueError("log_probability must map a batch [W, ndim] to [W]")
    return out


def gradient(log_probability, theta, h):
    """(value, central-difference gradient) from one batch of 2n + 1 points."""
    n = theta.size
    pts = np.repeat(theta[None, :], 2 * n + 1, axis=0)
    for i in range(n):
        pts[1 + 2 * i, i] += h[i]
        pts[2 + 2 * i, i] -= h[i]
    f = _batched(log_probability, pts)
    with np.errstate(invalid="ignore"):  # a stencil point outside the box gives -inf; the caller handles it
        return f[0], (f[1::2] - f[2::2]) / (2 * h)


def _hessian_points(theta, h):
    n = theta.size
    pts = []
    for i in range(n):
        for j in range(i, n):
            if i == j:
                for s in (+1, -1):
                    p = theta.copy(); p[i] += s * h[i]; pts.append(p)
            else:
                for si, sj in ((+1, +1), (+1, -1), (-1, +1), (-1, -1)):
                    p = theta.copy(); p[i] += si * h[i]; p[j] += sj * h[j]; pts.append(p)
    return np.array(pts)


def _hessian_from_values(f0, f, n, h):
    H = np.empty((n, n))
    k = 0
    for i in range(n):
        for j in range(i, n):
            if i == j:
                H[i, i] = (f[k] - 2 * f0 + f[k + 1]) / (h[i] * h[i]); k += 2
            else:
                H[i, j] = H[j, i] = (f[k] - f[k + 1] - f[k + 2] + f[k + 3]) / (4 * h[i] * h[j]); k += 4
    return H


def hessian(log_probability, theta, h):
    """Richardson-extrapolated central-difference Hessian; all 4 n^2 + 1 evaluations in one batch."""
    theta = np.asarray(theta, dtype=np.float64)
    n = theta.size
    p1, p2 = _hessian_points(theta, h), _hessian_points(theta, 2 * h)
    f = _batched(log_probability, np.vstack([theta[None, :], p1, p2]))
    f = np.where(np.isinf(f), -1e10, f)  # log_evidence.py:38-42
    H1 = _hessian_from_values(f[0], f[1:1 + len(p1)], n, h)
    H2 = _hessian_from_values(f[0], f[1 + len(p1):], n, 2 * h)
    return (4 * H1 - H2) / 3


def log_evidence(mc_samples, log_probs, log_probability, bounds, rel_step=1e-4, return_details=False):
    """Laplace ln Z.  ``log_probability``: batch callable; ``bounds``: [ndim, 2] as in the reference scripts."""
    mc_samples, log_probs = np.asarray(mc_samples, dtype=np.float64), np.asarray(log_probs, dtype=np.float64)
    bounds = np.asarray(bounds, dtype=np.float64)
    best = int(np.argmax(log_probs))
    x0 = mc_samples[best]
    n = x0.size
    h = rel_step * (bounds[:, 1] - bounds[:, 0])  # steps scaled to the prior box

    def fun_and_grad(theta):
        val, grad = gradient(log_probability, theta, h * 1e-2)
        if not np.isfinite(val) or not np.all(np.isfinite(grad)):
            return 1e10, np.zeros(n)  # log_evidence.py:20-24
        return -val, -grad

    res = minimize(fun_and_grad, x0=x0, jac=True, bounds=bounds, method="L-BFGS-B")
    if res.success and -res.fun >= log_probs[best]:
        theta_map, log_post_map = res.x, -res.fun
    else:  # log_evidence.py:30-33
        theta_map, log_post_map = x0, float(log_probs[best])
    # keep every stencil point strictly inside the box (outside it log P = -inf)
    room = np.minimum(theta_map - bounds[:, 0], bounds[:, 1] - theta_map)
    hh = np.minimum(h, room / 4.5)
    H = hessian(log_probability, theta_map, hh)
    neg_H = -H
    eig = np.linalg.eigvalsh(neg_H)
    if eig.min() <= 0:  # log_evidence.py:52-54
        neg_H = neg_H + (abs(eig.min()) + 1e-6 * np.max(np.abs(eig))) * np.eye(n)
    sign, logdet = np.linalg.slogdet(neg_H)
    ln_z = np.nan if sign <= 0 else log_post_map + 0.5 * n * np.log(2 * np.pi) - 0.5 * logdet
    if return_details:
        return ln_z, dict(theta_map=theta_map, log_post_map=log_post_map, hessian=H, n_batches=res.nfev + 1)
    return ln_z
