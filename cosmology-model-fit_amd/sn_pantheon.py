"""
Host-side mirror of the reference script ``sn/pantheon.py`` (Pantheon+ flat-LambdaCDM with a
peculiar-velocity step): the same module-level names — ``bounds``, ``chi_squared``,
``log_likelihood``, ``log_prior``, ``log_probability`` (+ the batch form
``log_probs_vectorized`` of bao/desi.py:100-106) — bound to a GPU engine.

    lk = sn_pantheon.PantheonLikelihood(z_cmb, z_hel, mb_vals, cov_matrix)
    sampler = emcee.EnsembleSampler(n_walkers, 4, lk.log_probs_vectorized, vectorize=True, moves=moves)

theta = (M, H0, Omega_m, v/100 km/s), sn/pantheon.py:68-75.
"""
import numpy as np
from scipy.linalg import cho_factor

from . import _lib as L
from .engine import C_KM_S, LikelihoodEngine, Param, solve_mode_of

# sn/pantheon.py:68-75
bounds = np.array(
    [
        (-20.0, -19.0),  # M
        (50.0, 90.0),  # H0
        (0.0, 0.7),  # Om
        (-3.0, 3.0),  # v x 100 km/s
    ]
)
H0_PRIOR = (1, 70.39, 1.80)  # TRGB prior, sn/pantheon.py:85
Z_TURN = 0.15  # sn/pantheon.py:46
N_GRID = 4000  # sn/pantheon.py:16


class PantheonLikelihood:
    def __init__(self, z_cmb, z_hel, mb_vals, cov_matrix=None, *, chol=None, device=0, devices=None, bounds=bounds,
                 h0_prior=H0_PRIOR, fde=L.CF_FDE_LCDM, step=None, fixed_mu=None, z_turn=Z_TURN, solve="auto", latency_mode=None,
                 probe_limit=0.0):
        """step: per-SN velocity weights instead of the +-1 Heaviside step (dipole fits, sn/pantheon_dipole.py:60-68:
        cos(angle) * attenuation * survey mask); fixed_mu: Cepheid distance moduli of calibrator hosts, NaN elsewhere
        (sn/pantheon_and_sh0es.py:63-69)."""
        z_cmb = np.asarray(z_cmb, dtype=np.float64)
        if chol is None:
            chol = cho_factor(cov_matrix, lower=True)[0]  # sn/pantheon.py:14
        self.bounds = np.asarray(bounds, dtype=np.float64)
        self.h0_prior = tuple(h0_prior) if h0_prior else None
        self.normalization = -np.sum(np.log(self.bounds[:, 1] - self.bounds[:, 0]))  # sn/pantheon.py:77
        self.z_cmb, self.z_hel, self.mb_vals = z_cmb, np.asarray(z_hel, float), np.asarray(mb_vals, float)
        self.z_max = float(np.max(z_cmb) + 0.1)  # sn/pantheon.py:16
        self.step = None if step is None else np.asarray(step, dtype=np.float64)
        self.z_turn = z_turn
        self.engine = LikelihoodEngine(
            ndim=4, z_max=self.z_max, n_grid=N_GRID, fde=fde,
            params=dict(offset=Param(0), H0=Param(1), Om=Param(2), v=Param(3)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mb_vals, chol=chol, z_turn=z_turn, step=step, fixed_mu=fixed_mu),
            bounds=self.bounds, gauss=[h0_prior] if h0_prior else [], device=device, devices=devices,
            solve_mode=solve_mode_of(solve, latency_mode), probe_limit=probe_limit,
        )

    # -- reference names ----------------------------------------------------------------------
    def chi_squared(self, params):
        return self.engine.chi_squared(params)

    def log_likelihood(self, params):
        return self.engine.log_likelihood(params)

    def log_probability(self, params):
        return self.engine.log_probability(params)

    def log_probs_vectorized(self, batch):
        """emcee ``vectorize=True`` / nautilus ``vectorized=True`` callback: [W, 4] -> float64[W]."""
        return self.engine.log_probability(np.atleast_2d(batch))

    def log_prior(self, params):
        """Host-side (trivial) restatement of sn/pantheon.py:80-85 for callers that want it alone."""
        p = np.asarray(params, dtype=np.float64)
        if not np.all((self.bounds[:, 0] < p) & (p < self.bounds[:, 1])):
            return -np.inf
        if self.h0_prior is None:
            return self.normalization
        idx, mean, sigma = self.h0_prior
        return self.normalization - 0.5 * (p[idx] - mean) ** 2 / sigma**2

    # -- accessors the post-fit plots use, with the reference's own signatures (sn/pantheon.py:34-54,152-155) --------
    def DM_z(self, params, z=None):
        """``DM_z(params, z)``: comoving distance at redshifts z (default: the sample's z_cmb), sn/pantheon.py:34-40."""
        if z is None:
            return self.engine.parts(params)["dm"][0]
        return self.engine.DM_z(params, z)

    def mu_theory(self, DM):
        """``mu_theory(DM)``: 25 + 5 log10((1 + z_hel) DM), sn/pantheon.py:52-54.  (A parameter vector instead of a
        distance array is accepted too: then DM = DM_z(params, z_cmb).)"""
        DM = np.asarray(DM, dtype=np.float64)
        if DM.shape != self.z_hel.shape:
            DM = self.DM_z(DM)
        return 25.0 + 5 * np.log10((1.0 + self.z_hel) * DM)

    def mu_corr(self, params, DM_obs=None):
        """``mu_corr(params, DM_obs)``: 5 log10(DM_z(params, z_cosmo) / DM_obs) with the velocity step, sn/pantheon.py:43-49.
        Without DM_obs the kernel's own accessor path returns it for DM_obs = DM_z(params, z_cmb)."""
        if DM_obs is None:
            return self.engine.parts(params)["mu_corr"][0]
        p = np.asarray(params, dtype=np.float64)
        step = self.step if self.step is not None else np.where(self.z_cmb <= self.z_turn, 1.0, -1.0)
        v_km_s = 100 * p[3] * step
        z_cosmo = -1.0 + (1.0 + self.z_cmb) / (1.0 + v_km_s / C_KM_S)
        return 5.0 * np.log10(self.engine.DM_z(p, z_cosmo) / np.asarray(DM_obs, dtype=np.float64))
