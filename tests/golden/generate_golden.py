#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE's own functions.

Run from the repo root, in the build container only (needs /root/reference):

    python tests/golden/generate_golden.py [case ...]

Each case runs in its own subprocess (``cmb.set_HZ`` is a process-wide global in the reference, and
the SN scripts need a data module injected before import).  The reference is imported from
/root/reference with cwd=/root/reference (its loaders use relative paths) and with
tests/golden/_numba_stub first on sys.path (identity ``njit``; numba is not installed).
Nothing from the reference is copied: the outputs are numeric fixtures (inputs + expected values).

Large SN covariances are absent from the reference snapshot (.MISSING_LARGE_BLOBS), so SN cases
inject ``yYYYY.../data.py`` replacements that return the REAL redshift / magnitude columns and a
seeded synthetic SPD covariance  C = diag(sigma^2) + A A^T,  A = 0.01 * N(0,1)[N,40]  (seed 0).
"""
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _enter_reference():
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(HERE, "_numba_stub"), REF]
    os.chdir(REF)


def synthetic_cov(sigma, seed=0, rank=40, amp=0.01):
    """Seeded SPD covariance; regenerated (never stored) by tests from the same recipe."""
    rng = np.random.default_rng(seed)
    A = amp * rng.standard_normal((sigma.size, rank))
    return np.diag(sigma**2) + A @ A.T


def theta_batch(bounds, n_in, rng):
    """In-box random rows + rows on / outside the box edges."""
    lo, hi = bounds[:, 0], bounds[:, 1]
    inside = rng.uniform(lo, hi, size=(n_in, len(lo)))
    mid = 0.5 * (lo + hi)
    edge = []
    for k in range(len(lo)):
        for val in (lo[k], hi[k], lo[k] - 0.1 * (hi[k] - lo[k]), hi[k] + 0.1 * (hi[k] - lo[k])):
            row = mid.copy()
            row[k] = val
            edge.append(row)
    return np.vstack([inside, mid[None, :], np.array(edge)])


# ------------------------------------------------------------------------------------------
def case_interpolator():
    _enter_reference()
    import interpolator as ip
    import solve_triangular as st

    rng = np.random.default_rng(1)
    out = {}
    # uniform grid like the hot path + queries on nodes, between nodes and outside
    x = np.linspace(0, 2.36137, 400)
    y = np.cumsum(rng.uniform(0.5, 1.5, x.size))
    yp = rng.uniform(0.2, 2.0, x.size)
    xq = np.concatenate([rng.uniform(-0.1, 2.5, 200), x[::37], [x[0], x[-1], -0.3, 3.0]])
    out["h_x"], out["h_y"], out["h_yp"], out["h_xq"] = x, y, yp, xq
    out["h_out"] = ip.interp_hermite(xq, x, y, yp)
    # non-uniform grid, non-monotone data: exercises every PCHIP slope branch
    x2 = np.sort(rng.uniform(0, 10, 60))
    y2 = np.sin(x2) + 0.3 * rng.standard_normal(60)
    y2[10:13] = y2[10]  # flat run -> delta == 0
    y2[0], y2[1], y2[2] = 0.0, 1.0, 0.5  # end-point sign change / overshoot
    y2[-1], y2[-2], y2[-3] = 2.0, 0.1, 0.3
    xq2 = np.concatenate([rng.uniform(-1, 11, 150), x2[::7]])
    out["p_x"], out["p_y"], out["p_xq"] = x2, y2, xq2
    out["p_slopes"] = ip._pchip_slopes(x2, y2)
    out["p_out"] = ip.interp_pchip(xq2, x2, y2)
    for name, yy in (("p3", np.array([1.0, 1.0, 2.0, 4.0, 3.0])), ("p4", np.array([0.0, 3.0, 2.9, 2.8, -5.0]))):
        xx = np.array([0.0, 1.0, 1.5, 4.0, 4.2])
        out[name + "_x"], out[name + "_y"] = xx, yy
        out[name + "_slopes"] = ip._pchip_slopes(xx, yy)
    # monotone smooth data (the DH grid case)
    x3 = np.linspace(0, 2.5, 300)
    y3 = 299792.458 / (70.0 * np.sqrt(0.3 * (1 + x3) ** 3 + 0.7))
    xq3 = rng.uniform(-0.05, 2.6, 64)
    out["m_x"], out["m_y"], out["m_xq"] = x3, y3, xq3
    out["m_out"] = ip.interp_pchip(xq3, x3, y3)
    # forward substitution with GARBAGE above the diagonal (cho_factor semantics)
    n = 129
    M = rng.standard_normal((n, n))
    C = M @ M.T + n * np.eye(n)
    L = np.linalg.cholesky(C) + np.triu(rng.standard_normal((n, n)), 1) * 7.0
    b = rng.standard_normal((5, n))
    out["t_L"], out["t_b"] = L, b
    out["t_out"] = np.array([st.solve_triangular(L, bb) for bb in b])
    np.savez_compressed(os.path.join(HERE, "interpolator.npz"), **out)
    print("interpolator.npz", {k: v.shape for k, v in out.items()})


def _inject_pantheon():
    import pandas as pd

    df = pd.read_csv(os.path.join(REF, "y2022pantheonSHOES/raw-data/distances.txt"), sep=" ")
    sel = df["zHD"].to_numpy(np.float64) > 0.01  # y2022pantheonSHOES/data.py:25
    z = df["zHD"].to_numpy(np.float64)[sel]
    zh = df["zHEL"].to_numpy(np.float64)[sel]
    mb = df["m_b_corr"].to_numpy(np.float64)[sel]
    sig = df["m_b_corr_err_DIAG"].to_numpy(np.float64)[sel]
    cov = synthetic_cov(sig)
    pkg = types.ModuleType("y2022pantheonSHOES")
    pkg.__path__ = []
    mod = types.ModuleType("y2022pantheonSHOES.data")
    mod.get_data = lambda: ("Pantheon+ (synthetic cov)", z, zh, mb, cov)
    sys.modules["y2022pantheonSHOES"] = pkg
    sys.modules["y2022pantheonSHOES.data"] = mod
    return z, zh, mb, sig


def case_sn_pantheon():
    _enter_reference()
    z, zh, mb, sig = _inject_pantheon()
    import sn.pantheon as m

    rng = np.random.default_rng(2)
    thetas = theta_batch(m.bounds, 24, rng)
    chi2 = np.array([m.chi_squared(t) for t in thetas])
    logp = np.array([m.log_probability(t) for t in thetas])
    logl = np.array([m.log_likelihood(t) for t in thetas])
    out = dict(z_cmb=z, z_hel=zh, obs=mb, sigma=sig, bounds=m.bounds, thetas=thetas,
               chi2=chi2, logp=logp, logl=logl, z_max=np.float64(m.z_grid[-1]),
               z_grid_sub=m.z_grid[::250], dz_sub=m.dz[::250])
    # intermediates for three walkers (incl. v != 0)
    for k, t in enumerate(thetas[:3]):
        DM = m.DM_z(t, m.z_cmb)
        out[f"dm_{k}"] = DM
        out[f"mucorr_{k}"] = m.mu_corr(t, DM)
        out[f"muth_{k}"] = m.mu_theory(DM)
        out[f"delta_{k}"] = m.mb_vals - t[0] - m.mu_corr(t, DM) - m.mu_theory(DM)
        dh = m.c / m.H_z(t)
        cum = np.zeros(m.z_grid.size)
        cum[1:] = np.cumsum(((dh[:-1] + dh[1:]) / 2) * m.dz)
        out[f"dh_sub_{k}"] = dh[::250]
        out[f"cum_sub_{k}"] = cum[::250]
    np.savez_compressed(os.path.join(HERE, "sn_pantheon.npz"), **out)
    print("sn_pantheon.npz chi2[:4] =", chi2[:4], "logp[-4:] =", logp[-4:])


CASES = {
    "interpolator": case_interpolator,
    "sn_pantheon": case_sn_pantheon,
}

if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        CASES[sys.argv[2]]()
    else:
        for name in sys.argv[1:] or list(CASES):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], check=True)
