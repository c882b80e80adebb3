"""
Host-side mirrors of the reference's joint-likelihood scripts: the same quantities under the same names
(``chi_squared``, ``log_likelihood``, ``log_probability``, ``bounds``, ``bao_theory`` ...) bound to a GPU engine.
The caller passes the data arrays the reference's own loaders return (``get_data()`` tuples); nothing is loaded
from disk here.

    bao/desi.py                         -> DesiBao              (BAO only, late-time flat, thawing w0, fixed r_d)
    bao/desi_cmb.py                     -> DesiCmb              (BAO + early-LCDM theta* compression, thawing w0)
    bao/desi_fs_lya_cmb.py              -> DesiFsLyaCmb         (BAO FS+Lya incl. F_AP + Planck/ACT, CPL, w0+wa wall)
    bao/desi_cmb_des5y.py               -> DesiCmbDes5y         (SN + BAO + CMB, BASELINE config 3 as shipped)
    bao/desi_des5y_bbn_theta_star.py    -> DesiDes5yBbnThetaStar(SN + BAO + l_A + BBN prior, BASELINE config 5)
    sn/union3_1.py                      -> SnUnion3             (22 binned distances, explicit inverse covariance)
    bao/desi_union3_cc_theta_star.py    -> DesiUnion3CcThetaStar(Union3 + BAO + l_A + cosmic chronometers with f_cc)
    sn/pantheon_dipole.py, sn/pantheon_and_sh0es.py -> sn_pantheon.PantheonLikelihood(step=..., fixed_mu=...)
"""
import numpy as np
from scipy.linalg import cho_factor

from . import _lib as L
from . import cmb_data
from .engine import C_KM_S, LikelihoodEngine, Param, solve_mode_of

N_GRID = 4000
QTY_MAP = {"DV_over_rs": 0, "DM_over_rs": 1, "DH_over_rs": 2, "F_AP": 3}  # bao/desi_cmb_des5y.py:69-78


def bao_arrays(bao_data, cov):
    """(z, value, qty codes, inverse covariance) from the reference's structured BAO array + covariance."""
    qty = np.array([QTY_MAP[str(q)] for q in bao_data["quantity"]], dtype=np.int32)
    return np.asarray(bao_data["z"], float), np.asarray(bao_data["value"], float), qty, np.linalg.inv(cov)


def _physical(comp):
    return {k: comp[k] for k in ("or_h2", "omnu_h2", "o_gamma_h2", "nu_m0", "nu_rho0", "nu_qs_sq", "nu_ws")}


class _Base:
    bounds = None

    def chi_squared(self, params):
        return self.engine.chi_squared(params)

    def log_likelihood(self, params):
        return self.engine.log_likelihood(params)

    def log_probability(self, params):
        return self.engine.log_probability(params)

    def log_probs_vectorized(self, batch):
        return self.engine.log_probability(np.atleast_2d(batch))

    log_probability_vect = log_probs_vectorized  # bao/desi_cmb.py:137

    def bao_theory(self, *args):
        """``bao_theory(params)``: the theory vector at the data points.  With the scripts' own signatures
        ``bao_theory(z, qty, params)`` (bao/desi.py:38) / ``bao_theory(z, qty, params, DM_interp)`` (bao/desi_cmb_des5y.py:82):
        the same quantities at ARBITRARY redshifts -- what the post-fit block passes to ``plot_bao_predictions`` as
        ``lambda z, qty: bao_theory(z, qty, best_fit)`` (bao/desi.py:204-211; DM_interp is implied by params and ignored)."""
        if len(args) == 1:
            return self.engine.parts(args[0])["bao_theory"][0]
        if len(args) in (3, 4):
            z, qty, params = args[:3]
            return self.engine.bao_theory_at(params, z, qty)
        raise TypeError("bao_theory(params) or bao_theory(z, qty, params[, DM_interp])")

    def H_z(self, z, params):
        """``H_z(z, params)`` of the scripts (km/s/Mpc) at arbitrary redshifts: the curve of ``plot_cc_predictions``
        (``lambda z: H_z(z, best_fit)``, ohd/cc.py:95-96, bao/desi_cc.py:193-194)."""
        return self.engine.H_z(params, z)

    def fs8_theory(self, *args):
        """``fs8_theory(params)``: f sigma_8 at the data points.  ``fs8_theory(a, params)`` -- the scripts' own signature
        (fs8/fs8_cmb.py:132, bao/desi_cmb_union3_fs8.py:172, ohd/cc_fs8.py:90) -- at ARBITRARY scale factors: the smooth curve of
        the post-fit block (``lambda z: fs8_theory(1 / (1 + z), best_fit)``, fs8/plot_predictions.py:7-32)."""
        if len(args) == 1:
            return self.engine.parts(args[0])["fs8_theory"][0]
        if len(args) == 2:
            return self.engine.fs8_theory_at(args[1], 1.0 / np.asarray(args[0], dtype=np.float64) - 1.0)
        raise TypeError("fs8_theory(params) or fs8_theory(a, params)")

    def cmb_distances(self, params):
        return self.engine.parts(params)["cmb_vector"][0]

    def DM_z(self, z, params):
        """``DM_z(z, params)`` of the joint scripts (bao/desi_cmb_des5y.py:60-66 + interp_hermite): comoving distance at
        arbitrary redshifts from the walker's table (GPU) and the GPU Hermite operator."""
        return self.engine.DM_z(params, z)


class DesiBao(_Base):
    """bao/desi.py: theta = (h, Om, w0); bounds bao/desi.py:69-75; r_d = 147.09 Mpc fixed (:10)."""
    bounds = np.array([(0.50, 0.80), (0.1, 0.5), (-1.0, 0.0)])

    def __init__(self, z, val, qty, inv_cov, *, rd=147.09, device=0, devices=None, bounds=None):
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z) + 0.1)  # bao/desi.py:19
        self.engine = LikelihoodEngine(
            ndim=3, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_THAWING,
            params=dict(H0=Param(0, scale=100.0), Om=Param(1), w0=Param(2), rd=Param(fixed=rd)),
            bao=dict(z=z, val=val, qty=qty, inv_cov=inv_cov), bounds=self.bounds, device=device, devices=devices)


class DesiCmb(_Base):
    """bao/desi_cmb.py: theta = (H0, wb, wc, w0); exact D_H, r_drag fit, (theta*, wb, wm) compression."""
    bounds = np.array([(50.0, 80.0), (0.020, 0.024), (0.05, 0.30), (-1.0, 0.0)])

    def __init__(self, z, val, qty, inv_cov, *, comp=None, device=0, devices=None, bounds=None):
        comp = cmb_data.EARLY_LCDM if comp is None else comp
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z) + 0.1)
        self.engine = LikelihoodEngine(
            ndim=4, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_THAWING,
            params=dict(H0=Param(0), obh2=Param(1), och2=Param(2), w0=Param(3)),
            bao=dict(z=z, val=val, qty=qty, inv_cov=inv_cov, dh_exact=True, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), bounds=self.bounds, device=device, devices=devices)


class DesiFsLyaCmb(_Base):
    """bao/desi_fs_lya_cmb.py: theta = (H0, wb, wc, w0, wa); CPL; log L = -1e8 when w0 + wa >= 0 (:118-121)."""

    def __init__(self, z, val, qty, inv_cov, *, comp=None, device=0, devices=None):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        self.z_max = float(np.max(z) + 0.1)
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_CPL,
            params=dict(H0=Param(0), obh2=Param(1), och2=Param(2), w0=Param(3), wa=Param(4)),
            bao=dict(z=z, val=val, qty=qty, inv_cov=inv_cov, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), cpl_wall=True, device=device, devices=devices)


FDE_BY_NAME = {"lcdm": L.CF_FDE_LCDM, "wcdm": L.CF_FDE_WCDM, "thawing": L.CF_FDE_THAWING, "cpl": L.CF_FDE_CPL}


class DesiCmbDes5y(_Base):
    """bao/desi_cmb_des5y.py: theta = (dM, H0, wb, wc, v); SN with velocity step at z = 0.10563 (:105), BAO with
    PCHIP D_H and F_AP, Planck+ACT (R, l_A, wb); dark energy = Lambda as shipped (:46).
    ``fde``: the dark-energy lines the author toggles by commenting (:26-31) -- "lcdm" (as shipped), "wcdm" and "thawing"
    (theta gains w0), "cpl" (theta gains w0, wa: BASELINE configs[2] as worded, w0waCDM); the dark-energy density then
    multiplies Ode in Ez exactly as in bao/desi_fs_lya_cmb.py:19-22,40-49; ``cpl_wall`` adds that script's
    w0 + wa >= 0 -> -1e8 wall (:118-121).
    bao/desi_cmb_pantheon.py is the same likelihood on Pantheon+ with the step at z = 0.15 (:102) and D_H = c / H
    exactly (:62-63): ``DesiCmbDes5y(..., z_turn=0.15, dh_exact=True)`` (alias DesiCmbPantheon)."""

    def __init__(self, z_cmb, z_hel, mu_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, *, chol=None, comp=None,
                 device=0, devices=None, solve="auto", latency_mode=None, z_turn=0.10563, dh_exact=False, fde="lcdm",
                 cpl_wall=False):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        if chol is None:
            chol = cho_factor(cov_sn, lower=True)[0]  # bao/desi_cmb_des5y.py:17
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z)) + 0.1)  # :20
        params = dict(offset=Param(0), H0=Param(1), obh2=Param(2), och2=Param(3), v=Param(4))
        if fde != "lcdm":
            params["w0"] = Param(5)
        if fde == "cpl":
            params["wa"] = Param(6)
        self.ndim = len(params)
        self.engine = LikelihoodEngine(
            ndim=self.ndim, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=FDE_BY_NAME[fde],
            params=params,
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_values, chol=chol, z_turn=z_turn),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, rd_fit=comp["rd_fit"], dh_exact=dh_exact),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), device=device, devices=devices, cpl_wall=cpl_wall,
            solve_mode=solve_mode_of(solve, latency_mode))


class DesiCmbPantheon(DesiCmbDes5y):
    """bao/desi_cmb_pantheon.py: theta = (M, H0, wb, wc, v)."""

    def __init__(self, z_cmb, z_hel, mb_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, **kw):
        kw.setdefault("z_turn", 0.15)
        kw.setdefault("dh_exact", True)
        super().__init__(z_cmb, z_hel, mb_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, **kw)


class DesiCmbDes5yH0Trgb(_Base):
    """bao/desi_cmb_des5y_H0trgb.py: theta = (dM, H0, wb, wc, v); SN with velocity step at z = 0.11 (:110), DESI DR2 BAO
    and the single 6dF D_V point as one block-diagonal BAO block (:139-144), exact D_H (:63-64), Planck+ACT
    (R, l_A, wb), and the TRGB chi^2 term ((H0 - 70.39) / 1.80)^2 (:149); dark energy = Lambda as shipped (:52)."""
    H0_TRGB = (70.39, 1.80)

    def __init__(self, z_cmb, z_hel, mu_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, sixdf_z, sixdf_val, sixdf_qty,
                 sixdf_inv_cov, *, chol=None, comp=None, device=0, devices=None, solve="auto"):
        from scipy.linalg import block_diag

        comp = cmb_data.PLANCK_ACT if comp is None else comp
        if chol is None:
            chol = cho_factor(cov_sn, lower=True)[0]  # :18
        z_all = np.concatenate([np.asarray(bao_z, float), np.atleast_1d(np.asarray(sixdf_z, float))])
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z)) + 0.1)  # :22 (the 6dF point lies below the DESI ones)
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_LCDM,
            params=dict(offset=Param(0), H0=Param(1), obh2=Param(2), och2=Param(3), v=Param(4)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_values, chol=chol, z_turn=0.11),
            bao=dict(z=z_all, val=np.concatenate([np.asarray(bao_val, float), np.atleast_1d(np.asarray(sixdf_val, float))]),
                     qty=np.concatenate([np.asarray(bao_qty), np.atleast_1d(np.asarray(sixdf_qty))]),
                     inv_cov=block_diag(np.asarray(bao_inv_cov, float), np.atleast_2d(np.asarray(sixdf_inv_cov, float))),
                     dh_exact=True, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), chi2_gauss=[(1, self.H0_TRGB[0], self.H0_TRGB[1])], device=device, devices=devices,
            solve_mode=solve_mode_of(solve))


class SnCmb(_Base):
    """sn/pantheon_cmb.py and sn/des5y_cmb.py: theta = (M or dM, H0, wb, wc, v); SN with velocity step at z_turn (0.15,
    sn/pantheon_cmb.py:65; 0.11, sn/des5y_cmb.py:71) + Planck+ACT (R, l_A, wb), physical densities with Lambda as shipped;
    box prior of sn/pantheon_cmb.py:91-99 when `bounds` is given (the DES script samples with nautilus: log L only)."""
    PANTHEON_BOUNDS = np.array([(-20.0, -19.0), (60.0, 75.0), (0.010, 0.030), (0.010, 0.25), (-2.5, 2.5)])

    def __init__(self, z_cmb, z_hel, obs, cov_sn, *, z_turn, chol=None, comp=None, bounds=None, device=0, devices=None, solve="auto"):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        if chol is None:
            chol = cho_factor(cov_sn, lower=True)[0]
        self.bounds = None if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z_cmb) + 0.1)
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_LCDM,
            params=dict(offset=Param(0), H0=Param(1), obh2=Param(2), och2=Param(3), v=Param(4)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=obs, chol=chol, z_turn=z_turn),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), bounds=self.bounds, device=device, devices=devices, solve_mode=solve_mode_of(solve))


class DesiDes5yBbnThetaStar(_Base):
    """bao/desi_des5y_bbn_theta_star.py: theta = (dM, H0, wb, wc, w0); no velocity step, exact D_H, l_A only
    (delta^2 / covariance[1,1], :110-111), BBN prior on wb (:139); bounds :122-130."""
    bounds = np.array([(-0.5, 0.5), (50.0, 90.0), (0.010, 0.030), (0.05, 0.30), (-1.0, -1 / 3)])

    def __init__(self, z_cmb, z_hel, mu_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, *, chol=None, comp=None,
                 bbn=cmb_data.BBN_SCHONEBERG, device=0, devices=None, bounds=None):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        if chol is None:
            chol = cho_factor(cov_sn, lower=True)[0]
        inv = np.zeros((3, 3))
        inv[1, 1] = 1.0 / comp["cmb_cov"][1, 1]
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z)) + 0.1)
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_THAWING,
            params=dict(offset=Param(0), H0=Param(1), obh2=Param(2), och2=Param(3), w0=Param(4)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_values, chol=chol),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=True, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=2, prior=comp["cmb_prior"], inv_cov=inv, zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), bounds=self.bounds, gauss=[(2, bbn[0], bbn[1])], device=device, devices=devices)


class SnUnion3(_Base):
    """sn/union3_1.py: theta = (dM, Om, v); H0 fixed to 70 (:11), velocity step at z = 0.2 (:40).  The reference
    multiplies by the explicit inverse covariance (:57); the engine solves with its Cholesky factor (same chi^2)."""

    PRIOR_BOX = np.array([(-1.0, 1.0), (0.1, 0.7), (-9.0, 9.0)])  # the nautilus prior of main() (:77-79)

    def __init__(self, z_cmb, z_hel, mu_vals, cov_matrix, *, H0=70.0, bounds=None, device=0, devices=None):
        self.z_max = float(np.max(z_cmb) + 0.1)
        self.bounds = None if bounds is None else np.asarray(bounds, float)
        self.engine = LikelihoodEngine(
            ndim=3, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_LCDM,
            params=dict(offset=Param(0), H0=Param(fixed=H0), Om=Param(1), v=Param(2)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_vals, chol=np.linalg.cholesky(cov_matrix), z_turn=0.2),
            bounds=self.bounds, device=device, devices=devices)


class CcSn(_Base):
    """ohd/cc_des5y.py: theta = (f_cc, dM, H0, Om, w0); SN without a velocity step + cosmic chronometers with the
    error-rescale parameter f_cc and the Gaussian normalisation in log L (:72-96), late-time flat wCDM (:24-35),
    box prior (:59-68).  (ohd/cc_pantheon.py and ohd/cc_union3.py: the same blocks on the other SN sets.)"""
    bounds = np.array([(0.2, 3.0), (-0.5, 0.5), (50.0, 85.0), (0.05, 0.6), (-1.0, -1.0 / 3)])

    def __init__(self, z_cmb, z_hel, mu_values, cov_sn, z_cc, H_cc, cov_cc, *, chol=None, fde=L.CF_FDE_WCDM, bounds=None,
                 device=0, devices=None, solve="auto"):
        if chol is None:
            chol = cho_factor(cov_sn, lower=True)[0]
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z_cmb) + 0.1)  # :18
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_LATE_FLAT, fde=fde,
            params=dict(fcc=Param(0), offset=Param(1), H0=Param(2), Om=Param(3), w0=Param(4)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_values, chol=chol),
            cc=dict(z=z_cc, h=H_cc, inv_cov=np.linalg.inv(cov_cc), logdet=np.linalg.slogdet(cov_cc)[1]),
            bounds=self.bounds, device=device, devices=devices, solve_mode=solve_mode_of(solve))


class DesiUnion3CcThetaStar(_Base):
    """bao/desi_union3_cc_theta_star.py: theta = (f_cc, dM, H0, wb, wc, v).  Union3.1 SN (explicit inverse in the
    reference), DESI BAO with exact D_H, l_A only, cosmic chronometers: chi2_cc * f_cc^2 and the Gaussian
    normalisation with rescaled errors in log L (:129-139).  nautilus vectorized callback: ``log_likelihood``."""

    def __init__(self, z_cmb, z_hel, mu_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, z_cc, H_cc, cov_cc, *,
                 comp=None, device=0, devices=None):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        inv = np.zeros((3, 3))
        inv[1, 1] = 1.0 / comp["cmb_cov"][1, 1]
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z)) + 0.1)  # :24
        self.engine = LikelihoodEngine(
            ndim=6, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_LCDM,
            params=dict(fcc=Param(0), offset=Param(1), H0=Param(2), obh2=Param(3), och2=Param(4), v=Param(5)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_values, chol=np.linalg.cholesky(cov_sn), z_turn=0.2),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=True, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=2, prior=comp["cmb_prior"], inv_cov=inv, zstar_fit=comp["zstar_fit"]),
            cc=dict(z=z_cc, h=H_cc, inv_cov=np.linalg.inv(cov_cc), logdet=np.linalg.slogdet(cov_cc)[1]),
            physical=_physical(comp), device=device, devices=devices)


class DesiOmh2(_Base):
    """bao/desi_omh2.py: theta = (rd, H0, omega_m, w0).  BAO only; r_d free; the matter density is sampled as
    omega_m = Omega_m h^2 (:18-20, Planck prior on it applied by nautilus); thawing dark energy; D_H = c / H exactly.
    bao/desi_union3_omh2.py / desi_des5y_omh2.py add an SN block: ``DesiSnRd(..., omh2=True)``."""

    def __init__(self, z, val, qty, inv_cov, *, device=0, devices=None, bounds=None):
        self.bounds = None if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z) + 0.1)  # :13
        self.engine = LikelihoodEngine(
            ndim=4, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_THAWING, om_mode=1,
            params=dict(rd=Param(0), H0=Param(1), Om=Param(2), w0=Param(3)),
            bao=dict(z=z, val=val, qty=qty, inv_cov=inv_cov, dh_exact=True), bounds=self.bounds, device=device, devices=devices)


class DesiSnRd(_Base):
    """bao/desi_des5y_rd.py: theta = (dM, rd, H0, Om, v).  SN (velocity step at z = 0.10563, :86) + DESI BAO with PCHIP D_H
    and a FREE sound horizon r_d (its Gaussian prior lives in the sampler, :114); flat LCDM.
    ``omh2=True``: theta[3] is omega_m = Omega_m h^2 (bao/desi_des5y_omh2.py:31-32); ``z_turn`` 0.15 / 0.2 and the matching
    data give bao/desi_pantheon_rd.py and bao/desi_union3_rd.py."""

    def __init__(self, z_cmb, z_hel, mu_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, *, chol=None, z_turn=0.10563,
                 omh2=False, dh_exact=False, fde="lcdm", device=0, devices=None, solve="auto", bounds=None):
        if chol is None:
            chol = cho_factor(cov_sn, lower=True)[0]  # :12
        self.bounds = None if bounds is None else np.asarray(bounds, float)
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z)) + 0.1)  # :17
        params = dict(offset=Param(0), rd=Param(1), H0=Param(2), Om=Param(3), v=Param(4))
        if fde != "lcdm":
            params["w0"] = Param(5)
        self.engine = LikelihoodEngine(
            ndim=len(params), z_max=self.z_max, n_grid=N_GRID, fde=FDE_BY_NAME[fde], om_mode=int(omh2), params=params,
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_values, chol=chol, z_turn=z_turn),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=dh_exact),
            bounds=self.bounds, device=device, devices=devices, solve_mode=solve_mode_of(solve))


class DesiCmbPantheonH0Trgb(_Base):
    """bao/desi_cmb_pantheon_H0trgb.py: theta = (M, H0, wb, wc, v_flow).  Pantheon+ magnitudes with the linearised bulk-flow
    term M_i = M + 100 v_flow (5 / ln 10) / (c z_i) instead of a velocity step (:102-106), DESI BAO (exact D_H, r_drag
    fit), Planck+ACT (R, l_A, wb) and the TRGB term ((H0 - 70.39) / 1.80)^2 in chi^2 (:123)."""
    H0_TRGB = (70.39, 1.80)

    def __init__(self, z_cmb, z_hel, mb_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, *, chol=None, comp=None,
                 device=0, devices=None, solve="auto"):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        if chol is None:
            chol = cho_factor(cov_sn, lower=True)[0]  # :16
        z_cmb = np.asarray(z_cmb, dtype=np.float64)
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z)) + 0.1)  # :19
        lin_coef = 100 * (5 / np.log(10)) / (C_KM_S * z_cmb)  # :104
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_LCDM,
            params=dict(offset=Param(0), H0=Param(1), obh2=Param(2), och2=Param(3), lin=Param(4)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mb_values, chol=chol, lin_coef=lin_coef),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=True, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), chi2_gauss=[(1, self.H0_TRGB[0], self.H0_TRGB[1])], device=device, devices=devices,
            solve_mode=solve_mode_of(solve))


class PantheonDipoleXyz(_Base):
    """sn/pantheon_dipole_xyz.py: theta = (M, H0, Om, vx, vy, vz).  Bulk-flow velocity VECTOR in equatorial cartesian
    coordinates: v_los,i = n_i . v with n_i from (RA, DEC) (:13-17), times the tanh attenuation and the survey mask
    (:50-57).  ``dirs`` [N, 3] = (nx, ny, nz), ``weights`` [N] = attenuation * survey_mask (see ``dipole_geometry``)."""

    def __init__(self, z_cmb, z_hel, mb_vals, cov_matrix, dirs, weights, *, chol=None, device=0, devices=None, solve="auto",
                 bounds=None):
        if chol is None:
            chol = cho_factor(cov_matrix, lower=True)[0]  # :9
        self.bounds = None if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z_cmb) + 0.1)  # :22
        self.engine = LikelihoodEngine(
            ndim=6, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_LCDM,
            params=dict(offset=Param(0), H0=Param(1), Om=Param(2), v=Param(3), v2=Param(4), v3=Param(5)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mb_vals, chol=chol, step=weights, dirs=dirs),
            bounds=self.bounds, device=device, devices=devices, solve_mode=solve_mode_of(solve))

    @staticmethod
    def dipole_geometry(z_cmb, ra_deg, dec_deg, survey_id, target_ids=(1, 5, 15, 50, 51, 56, 63, 150), z_c=0.10, dz=0.02):
        """(dirs, weights) from sky positions as sn/pantheon_dipole_xyz.py:12-20,51-53 builds them (host side, once)."""
        ra, dec = np.deg2rad(ra_deg), np.deg2rad(dec_deg)
        dirs = np.stack([np.cos(dec) * np.cos(ra), np.cos(dec) * np.sin(ra), np.sin(dec)], axis=1)
        att = 0.5 * (1.0 - np.tanh((np.asarray(z_cmb) - z_c) / dz))
        return dirs, att * np.isin(survey_id, list(target_ids)).astype(int)


# ---- growth-rate scripts ------------------------------------------------------------------------------------
def fs8_fiducial(z, h_dm_of_z):
    """(H D_M)_fid per datum: ``h_dm_of_z(k, z_k)`` evaluates H(z_k) D_M(z_k) in datum k's fiducial cosmology, as the scripts
    do once at import (fs8/fs8.py:101-108).  See ``Fs8.fiducial`` for the flat-LCDM form on the scripts' own grid."""
    return np.array([h_dm_of_z(k, zk) for k, zk in enumerate(np.asarray(z, dtype=np.float64))])


def _flat_lcdm_h_dm(z_grid, z, H0, Om):
    """H(z) D_M(z) of flat LCDM with the scripts' arithmetic: trapezoid on their 4000-node grid, Hermite to z (host side, run
    once per datum at set-up: fs8/fs8.py:101-108; interpolation by the GPU operator)."""
    from .interpolator import interp_hermite

    dh_grid = C_KM_S / (H0 * np.sqrt(Om * (1.0 + z_grid) ** 3 + (1.0 - Om)))
    cum = np.zeros(z_grid.size)
    cum[1:] = np.cumsum(((dh_grid[:-1] + dh_grid[1:]) / 2) * np.diff(z_grid))
    return H0 * np.sqrt(Om * (1.0 + z) ** 3 + (1.0 - Om)) * interp_hermite(np.array([z]), z_grid, cum, dh_grid)[0]


class Fs8(_Base):
    """fs8/fs8.py: theta = (Om, sigma8, w0, f_err); bounds :128-135.  Growth-rate data alone, late-time flat with thawing
    dark energy, H0 drops out (E(z) and c / E: the engine runs with H0 = 1); chi2 = f_err^2 delta C^-1 delta and
    log L = -0.5 (chi2 - 2 N ln f_err) (:116-125)."""
    bounds = np.array([(0.1, 0.6), (0.5, 1.0), (-1.0, 0.0), (0.2, 3.2)])
    A_INIT = 10**-2.15  # :79
    N_A = 1000  # a_span = np.logspace(-2.15, 0, 1000) (:79)

    def __init__(self, z, fs8_vals, cov_mat, omega_fid, *, fid=None, device=0, devices=None, bounds=None, steps=0, a_grid=None):
        """a_grid: points of the a-grid delta' is interpolated on as the script does (None: the script's 1000); 0 reads delta' off
        the integration directly."""
        z = np.asarray(z, dtype=np.float64)
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z) + 0.1)  # :18
        if fid is None:  # E(z) D_M(z) at each datum's fiducial Omega_m, w0 = -1 (:101-108)
            z_grid = np.linspace(0, self.z_max, num=N_GRID)
            fid = fs8_fiducial(z, lambda k, zk: _flat_lcdm_h_dm(z_grid, zk, 1.0, omega_fid[k]))
        self.fid = np.asarray(fid, dtype=np.float64)
        self.engine = LikelihoodEngine(
            ndim=4, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_THAWING,
            params=dict(H0=Param(fixed=1.0), Om=Param(0), s8=Param(1), w0=Param(2), fs8err=Param(3)),
            fs8=dict(z=z, val=fs8_vals, inv_cov=np.linalg.inv(cov_mat), fid=self.fid, a_init=self.A_INIT, steps=steps,
                     a_grid=self.N_A if a_grid is None else a_grid),
            bounds=self.bounds, device=device, devices=devices)

    def fs8_theory(self, *args):
        """``fs8_theory(params)`` at the data points, or the script's own ``fs8_theory(a, Om, sigma8_0, w0)`` (fs8/fs8.py:84) at
        arbitrary scale factors."""
        if len(args) == 4:
            a, Om, s8, w0 = args
            return super().fs8_theory(a, np.array([Om, s8, w0, 1.0]))
        return super().fs8_theory(*args)


class DesiCmbUnion3Fs8(_Base):
    """bao/desi_cmb_union3_fs8.py: theta = (dM, H0, wb, wc, v, sigma8).  Union3.1 SN (explicit inverse in the reference; step
    at z = 0.2), DESI BAO with exact D_H, Planck+ACT (R, l_A, wb) and the growth-rate data with the physical-density
    H(z) (radiation and massive neutrinos enter dH/da, :127-145); the ODE starts at a = 10^-2.7 (:168)."""
    A_INIT = 10**-2.7
    N_A = 2500  # a_span = np.logspace(-2.7, 0, 2500) (bao/desi_cmb_union3_fs8.py:169)

    def __init__(self, z_cmb, z_hel, mu_values, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, fs8_z, fs8_vals, fs8_cov, fs8_fid, *,
                 comp=None, device=0, devices=None, steps=0):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z), np.max(fs8_z)) + 0.1)  # :26
        self.engine = LikelihoodEngine(
            ndim=6, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_LCDM,
            params=dict(offset=Param(0), H0=Param(1), obh2=Param(2), och2=Param(3), v=Param(4), s8=Param(5)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_values, chol=np.linalg.cholesky(cov_sn), z_turn=0.2),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=True, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            fs8=dict(z=fs8_z, val=fs8_vals, inv_cov=np.linalg.inv(fs8_cov), fid=fs8_fid, a_init=self.A_INIT, steps=steps, a_grid=self.N_A),
            physical=_physical(comp), device=device, devices=devices)


class CcFs8(_Base):
    """ohd/cc_fs8.py: theta = (H0, Om, sigma8, f_cc, f_fs8, w0).  Cosmic chronometers and growth-rate data, each with its own
    error-rescale factor; log L = -0.5 (chi2 - 2 N_cc ln f_cc - 2 N_fs8 ln f_fs8) without the constant of the Gaussian
    normalisation (:137-140: hence logdet = -N_cc ln 2 pi here); the ODE starts at a = 1 / (1 + 200) (:86-87); nautilus
    vectorised callback ``log_likelihood`` (:143-144)."""

    A_INIT = 1.0 / (1.0 + 200.0)  # max_z = 200, :86-87
    N_A = 1000  # a_span = np.logspace(log10 a_init, 0, 1000) (ohd/cc_fs8.py:87)

    def __init__(self, z_cc, H_cc, cov_cc, fs8_z, fs8_vals, fs8_cov, fs8_fid, *, device=0, devices=None, steps=0):
        z_top = float(max(np.max(fs8_z), np.max(z_cc)))
        self.z_max = z_top + 0.1  # :23-24
        self.engine = LikelihoodEngine(
            ndim=6, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_THAWING,
            params=dict(H0=Param(0), Om=Param(1), s8=Param(2), fcc=Param(3), fs8err=Param(4), w0=Param(5)),
            cc=dict(z=z_cc, h=H_cc, inv_cov=np.linalg.inv(cov_cc), logdet=-len(z_cc) * np.log(2 * np.pi)),
            fs8=dict(z=fs8_z, val=fs8_vals, inv_cov=np.linalg.inv(fs8_cov), fid=fs8_fid, a_init=self.A_INIT, steps=steps, a_grid=self.N_A),
            device=device, devices=devices)


class DesiUnion3ThetaStarSubset(_Base):
    """bao/desi_union3_omh2_theta_star.py: theta = (dM, H0, wb, wc, v).  Union3.1 SN + DESI BAO + a sub-vector of the early-LCDM
    CMB compression (theta*, omega_b, omega_m): ``components=(0, 2)`` keeps (theta*, omega_m) with the inverse of the 2 x 2
    sub-covariance (:17,110-112); (0, 1) and (1, 2) are the (theta*, omega_b) / (omega_b, omega_m) forms of
    bao/desi_pantheon_obh2_theta_star.py:23 and bao/desi_union3_obh2_theta_star.py:17,128.  The engine's 3 x 3 form takes the
    sub-inverse embedded in zeros: the omitted component drops out of the quadratic form exactly."""

    def __init__(self, z_cmb, z_hel, mu_vals, cov_sn, bao_z, bao_val, bao_qty, bao_inv_cov, *, components=(0, 2), comp=None,
                 z_turn=0.2, chol=None, device=0, devices=None, solve="auto"):
        comp = cmb_data.EARLY_LCDM if comp is None else comp
        idx = list(components)
        inv = np.zeros((3, 3))
        inv[np.ix_(idx, idx)] = np.linalg.inv(np.asarray(comp["cmb_cov"])[np.ix_(idx, idx)])
        if chol is None:
            chol = np.linalg.cholesky(cov_sn)
        self.z_max = float(max(np.max(z_cmb), np.max(bao_z)) + 0.1)  # :19
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_LCDM,
            params=dict(offset=Param(0), H0=Param(1), obh2=Param(2), och2=Param(3), v=Param(4)),
            sn=dict(z_cmb=z_cmb, z_hel=z_hel, obs=mu_vals, chol=chol, z_turn=z_turn),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=True, rd_fit=comp["rd_fit"]),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=inv, zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), device=device, devices=devices, solve_mode=solve_mode_of(solve))


class DesiBbn(_Base):
    """bao/desi_bbn.py: theta = (H0, Om, wb, w0); bounds :70-77.  BAO only in the late-time flat thawing model with PCHIP D_H; the
    sound horizon comes from the r_drag fit of the Planck compression with wm = Om h^2 (:46-60); the BBN omega_b measurement is a
    Gaussian term of the prior (:84-88)."""
    bounds = np.array([(55.0, 75.0), (0.17, 0.50), (0.016, 0.030), (-1.0, -1 / 3)])

    def __init__(self, z, val, qty, inv_cov, *, comp=None, bbn=cmb_data.BBN_SCHONEBERG, device=0, devices=None, bounds=None):
        comp = cmb_data.PLANCK if comp is None else comp
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z) + 0.1)  # :11
        self.engine = LikelihoodEngine(
            ndim=4, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_THAWING,
            params=dict(H0=Param(0), Om=Param(1), obh2=Param(2), w0=Param(3)),
            bao=dict(z=z, val=val, qty=qty, inv_cov=inv_cov, rd_fit=comp["rd_fit"], rd_wm_late=True),
            bounds=self.bounds, gauss=[(2, bbn[0], bbn[1])], device=device, devices=devices)


class DesiCc(_Base):
    """bao/desi_cc.py: theta = (f_cc, H0, r_d, Om, w0); bounds :89-97.  DESI BAO (D_H = c / H exactly, free r_d) + cosmic
    chronometers with the error-rescale factor f_cc and the Gaussian normalisation in log L (:107-110)."""
    bounds = np.array([(0.5, 2.5), (45.0, 90.0), (120.0, 175.0), (0.1, 0.7), (-1.0, 0.0)])

    def __init__(self, bao_z, bao_val, bao_qty, bao_inv_cov, z_cc, H_cc, cov_cc, *, device=0, devices=None, bounds=None):
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(bao_z) + 0.1)  # :19
        self.engine = LikelihoodEngine(
            ndim=5, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_THAWING,
            params=dict(fcc=Param(0), H0=Param(1), rd=Param(2), Om=Param(3), w0=Param(4)),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=True),
            cc=dict(z=z_cc, h=H_cc, inv_cov=np.linalg.inv(cov_cc), logdet=np.linalg.slogdet(cov_cc)[1]),
            bounds=self.bounds, device=device, devices=devices)


class Cc(_Base):
    """ohd/cc.py: theta = (H0, Om, f).  Cosmic chronometers alone, flat LCDM: chi2 = f^2 delta C^-1 delta and
    log L = -0.5 (chi2 + N ln 2 pi + logdet - 2 N ln f) (:22-35).  No distance table is needed (z_max is nominal)."""

    def __init__(self, z_cc, H_cc, cov_cc, *, device=0, devices=None, bounds=None):
        self.bounds = None if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z_cc) + 0.1)
        self.engine = LikelihoodEngine(
            ndim=3, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_LCDM,
            params=dict(H0=Param(0), Om=Param(1), fcc=Param(2)),
            cc=dict(z=z_cc, h=H_cc, inv_cov=np.linalg.inv(cov_cc), logdet=np.linalg.slogdet(cov_cc)[1]),
            bounds=self.bounds, device=device, devices=devices)


class Fs8Cmb(_Base):
    """fs8/fs8_cmb.py: theta = (H0, wb, wc, w0, sigma8, f_err); bounds :186-195.  Growth-rate data + Planck/ACT (R, l_A, wb) in the
    physical-density model with thawing dark energy; the growth ODE starts at a = 1 / 501 (:128-129); log L keeps
    -0.5 (N ln 2 pi + logdet) + N ln f_err (:19-21,181-183)."""
    bounds = np.array([(50, 80), (0.01, 0.035), (0.1, 0.35), (-1.0, 0.0), (0.5, 1.0), (0.2, 3.2)], dtype=float)
    A_INIT = 1.0 / 501.0
    N_A = 5000  # a_span = np.logspace(log10 a_init, 0, 5000) (fs8/fs8_cmb.py:129)

    def __init__(self, z, fs8_vals, cov_mat, fid, *, comp=None, device=0, devices=None, bounds=None, steps=0):
        comp = cmb_data.PLANCK_ACT if comp is None else comp
        z = np.asarray(z, dtype=np.float64)
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = float(np.max(z) + 0.1)  # :25
        norm_factor = len(z) * np.log(2 * np.pi) + np.linalg.slogdet(cov_mat)[1]  # :19-20
        self.engine = LikelihoodEngine(
            ndim=6, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_THAWING,
            params=dict(H0=Param(0), obh2=Param(1), och2=Param(2), w0=Param(3), s8=Param(4), fs8err=Param(5)),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            fs8=dict(z=z, val=fs8_vals, inv_cov=np.linalg.inv(cov_mat), fid=fid, a_init=self.A_INIT, steps=steps, a_grid=self.N_A),
            physical=_physical(comp), logl_const=-0.5 * norm_factor, bounds=self.bounds, device=device, devices=devices)


class DesiFsLyaCcFs8(_Base):
    """bao/desi_fs_lya_cc_fs8.py: theta = (H0, Om, sigma8, f_cc, f_fs8, r_d, w0).  DESI FS+Lya BAO (F_AP, D_H = c / H, free r_d),
    cosmic chronometers (f_cc) and growth-rate data (f_fs8), late-time flat thawing; the growth ODE starts at a = 1 / 201
    (:115-116); log L keeps both Gaussian normalisations (:183-192)."""
    A_INIT = 1.0 / 201.0
    N_A = 2000  # a_span = np.logspace(log10 a_init, 0, 2000) (bao/desi_fs_lya_cc_fs8.py:113)

    def __init__(self, bao_z, bao_val, bao_qty, bao_inv_cov, z_cc, H_cc, cov_cc, fs8_z, fs8_vals, fs8_cov, fs8_fid, *, device=0,
                 devices=None, steps=0):
        self.z_max = float(max(np.max(fs8_z), np.max(z_cc), np.max(bao_z)) + 0.1)  # :28-29
        norm_fs8 = len(fs8_z) * np.log(2 * np.pi) + np.linalg.slogdet(fs8_cov)[1]
        self.engine = LikelihoodEngine(
            ndim=7, z_max=self.z_max, n_grid=N_GRID, fde=L.CF_FDE_THAWING,
            params=dict(H0=Param(0), Om=Param(1), s8=Param(2), fcc=Param(3), fs8err=Param(4), rd=Param(5), w0=Param(6)),
            bao=dict(z=bao_z, val=bao_val, qty=bao_qty, inv_cov=bao_inv_cov, dh_exact=True),
            cc=dict(z=z_cc, h=H_cc, inv_cov=np.linalg.inv(cov_cc), logdet=np.linalg.slogdet(cov_cc)[1]),
            fs8=dict(z=fs8_z, val=fs8_vals, inv_cov=np.linalg.inv(fs8_cov), fid=fs8_fid, a_init=self.A_INIT, steps=steps, a_grid=self.N_A),
            logl_const=-0.5 * norm_fs8, device=device, devices=devices)


class CmbOnly(_Base):
    """cmb/cmb.py: theta = (H0, wb, wc); bounds :29-35.  The Planck+ACT (R, l_A, wb) compression alone.  ``log_likelihood`` and
    ``log_probability`` return (value, blobs) like the script (:45-70), blobs = (100 theta*, r_s(z*) in Mpc, D_M(z*) in Gpc, z*):
    the first three follow from the (R, l_A) the device computed (D_M* = R c / (100 sqrt(wm)), r* = pi D_M* / l_A), z* is the
    device's own evaluation of the fitting formula (``cf_eval_parts``)."""
    bounds = np.array([(60.0, 75.0), (0.020, 0.025), (0.05, 0.25)])

    def __init__(self, *, comp=None, device=0, devices=None, bounds=None):
        self.comp = comp = cmb_data.PLANCK_ACT if comp is None else comp
        self.bounds = self.bounds if bounds is None else np.asarray(bounds, float)
        self.z_max = 1.0  # no datum needs the distance table
        self.engine = LikelihoodEngine(
            ndim=3, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL, fde=L.CF_FDE_LCDM,
            params=dict(H0=Param(0), obh2=Param(1), och2=Param(2)),
            cmb=dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=comp["cmb_inv_cov"], zstar_fit=comp["zstar_fit"]),
            physical=_physical(comp), bounds=self.bounds, device=device, devices=devices)

    def blobs(self, params):
        th = np.atleast_2d(np.asarray(params, float))
        parts = self.engine.parts(th)
        vec = parts["cmb_vector"]
        wm = th[:, 1] + th[:, 2] + self.comp["omnu_h2"]
        dm_star = vec[:, 0] * C_KM_S / (100 * np.sqrt(wm))
        out = np.stack([100 * np.pi / vec[:, 1], np.pi * dm_star / vec[:, 1], dm_star / 1000, parts["z_star"]], axis=1)
        return out[0] if np.ndim(params) == 1 else out

    def log_likelihood(self, params):
        return self.engine.log_likelihood(params), self.blobs(params)

    def log_probability(self, params):
        """(log P, blobs); rows outside the box carry NaN blobs (the script returns np.empty(4) there, :66-67)."""
        lp = self.engine.log_probability(params)
        blobs = self.blobs(params)
        return lp, np.where(np.isfinite(np.asarray(lp))[..., None], blobs, np.nan)
