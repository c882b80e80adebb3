"""
Recipes: one entry per reference script that is a plain combination of the engine's blocks (SN / BAO / compressed CMB /
cosmic chronometers / growth rate) -- parameter order of theta, E(z) model, dark energy, the per-script conventions.  A
recipe is a description, not arithmetic: ``build(script, **data)`` turns it into a ``LikelihoodEngine`` descriptor, the
data arrays being what the script's own loaders return.

    lk = scripts.build("bao/desi_des5y_cc.py", sn=(z_cmb, z_hel, mu, cov_sn), bao=(bao_data, cov_bao), cc=(z_cc, H_cc, cov_cc))
    lk.log_probability(theta_batch)          # the script's log_probability, batched on the GPU

theta slot names: offset (M or dM), H0, Om, obh2, och2, w0, wa, v, rd, fcc, s8, fs8err (include/cosmofit.h cf_param_slot).
"""
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from . import cmb_data
from .engine import LikelihoodEngine, Param, solve_mode_of
from .likelihoods import FDE_BY_NAME, N_GRID, _Base, _physical, bao_arrays

H0_TRGB = (70.39, 1.80)  # TRGB H0: bao/desi_cmb_union3_H0trgb.py:140, bao/desi_des5y_H0trgb.py:111
RD_FIT_PLAIN = (1.0, 1.0) + cmb_data.RDRAG_A  # arXiv:2106.00428 eq. 8 without the (b, m) exponents: bao/desi_des5y_bbn.py:25-44


@dataclass
class Recipe:
    theta: Sequence[str]                 # slot names in the script's parameter order
    physical: bool = False               # E(z) from physical densities + radiation + massive neutrinos (needs `comp`)
    fde: str = "lcdm"                    # "lcdm" | "wcdm" | "thawing" | "cpl"
    comp: Optional[str] = None           # cmb_data compression: "PLANCK_ACT" | "EARLY_LCDM" | "PLANCK"
    omh2: bool = False                   # the "Om" slot holds omega_m = Omega_m h^2 (bao/desi_des5y_omh2.py:31-32)
    scale: dict = field(default_factory=dict)    # slot -> factor (sampler parameter h: {"H0": 100})
    fixed: dict = field(default_factory=dict)    # slot -> constant (r_d = 147.09)
    # SN: z_turn of the velocity step (None: no "v" slot or all weights +1), vel_mult: the multiplicative z_cosmo form
    sn: Optional[dict] = None
    # BAO: dh_exact (c / H at the datum, else PCHIP), rd: "free" (slot rd) | "fit" (omega_b + omega_c + omega_nu) |
    #      "fit_late" (Omega_m h^2) | "fit_late_plain" (same, b = m = 1)
    bao: Optional[dict] = None
    # CMB: components = indices of the compression vector that enter; sub = "cov" (inverse of the sub-covariance) |
    #      "inv" (sub-block of the full inverse)
    cmb: Optional[dict] = None
    cc: Optional[dict] = None            # f_inverse: f_cc multiplies the errors
    z_max_of: Sequence[str] = ("sn", "bao")      # blocks whose largest redshift sets the grid
    z_pad: float = 0.1
    bounds: Optional[Sequence] = None
    prior_normalised: bool = True
    gauss: Sequence = ()                 # (slot, mean, sigma) Gaussian terms of the log-prior
    chi2_gauss: Sequence = ()            # (slot, mean, sigma) Gaussian terms of chi^2
    cite: str = ""


_T5 = ("offset", "H0", "obh2", "och2", "v")
RECIPES = {
    "bao/desi_cmb_union3.py": Recipe(
        _T5, physical=True, comp="PLANCK_ACT", sn=dict(z_turn=0.2), bao=dict(dh_exact=False, rd="fit"), cmb=dict(),
        cite="theta :152-156; DESI FS+Lya + DES Y6 + 6dF as one block-diagonal BAO block (:20-21), PCHIP D_H (:83)"),
    "bao/desi_cmb_union3_H0trgb.py": Recipe(
        _T5, physical=True, comp="PLANCK_ACT", sn=dict(z_turn=0.2), bao=dict(dh_exact=True, rd="fit"), cmb=dict(),
        chi2_gauss=[("H0",) + H0_TRGB], cite="DESI + 6dF blocks (:126-134), TRGB term in chi^2 (:140)"),
    "bao/desi_des5y_H0trgb.py": Recipe(
        ("offset", "H0", "rd", "Om", "w0"), fde="thawing", sn=dict(z_turn=None), bao=dict(dh_exact=True, rd="free"),
        bounds=[(-0.5, 0.5), (56.0, 85.0), (120.0, 160.0), (0.1, 0.7), (-1.0, -1 / 3)], gauss=[("H0",) + H0_TRGB],
        cite="bounds :93-102, TRGB term in the prior (:109-112)"),
    "bao/desi_des5y_bbn.py": Recipe(
        ("H0", "Om", "obh2", "w0", "offset"), fde="thawing", sn=dict(z_turn=None), bao=dict(dh_exact=True, rd="fit_late_plain"),
        cite="r_drag(wb, Om h^2) :25-44,95-96; BAO through its Cholesky factor (:113-114)"),
    "bao/desi_union3_bbn.py": Recipe(
        ("H0", "Om", "obh2", "v", "offset"), sn=dict(z_turn=0.2), bao=dict(dh_exact=True, rd="fit_late_plain"),
        cite="DESI + DES Y6 BAO (:19-20), r_drag(wb, Om h^2) (:90,99)"),
    "bao/desi_bbn_theta_star.py": Recipe(
        ("H0", "obh2", "och2", "w0"), physical=True, fde="thawing", comp="PLANCK", bao=dict(dh_exact=True, rd="fit"),
        cmb=dict(components=(1,)), z_max_of=("bao",), cite="cmb.data_planck_compression (:5); l_A only (:93-96)"),
    "bao/desi_union3_bbn_theta_star.py": Recipe(
        _T5, physical=True, comp="PLANCK_ACT", sn=dict(z_turn=0.2), bao=dict(dh_exact=True, rd="fit"), cmb=dict(components=(1,)),
        cite="DESI FS+Lya + DES Y6 BAO (:19-20), l_A only (:132-134)"),
    "bao/desi_des5y_cc.py": Recipe(
        ("fcc", "offset", "H0", "rd", "Om", "v"), sn=dict(z_turn=0.10563), bao=dict(dh_exact=True, rd="free"), cc=dict(),
        bounds=[(0.5, 2.5), (-0.55, 0.55), (50.0, 80.0), (110.0, 175.0), (0.2, 0.7), (-4.5, 4.5)], cite="bounds :121-130"),
    "bao/desi_des5y_cc_theta_star.py": Recipe(
        ("fcc", "offset", "H0", "obh2", "och2", "w0"), physical=True, fde="thawing", comp="PLANCK_ACT", sn=dict(z_turn=None),
        bao=dict(dh_exact=True, rd="fit"), cmb=dict(components=(1,)), cc=dict(),
        bounds=[(0.5, 2.5), (-0.60, 0.60), (50.0, 85.0), (0.005, 0.035), (0.05, 0.30), (-1.0, -1 / 3)], cite="bounds :133-142"),
    "bao/desi_fs_lya.py": Recipe(
        ("H0", "Om", "w0"), fde="thawing", scale={"H0": 100.0}, fixed={"rd": 147.09}, bao=dict(dh_exact=False, rd="free"),
        z_max_of=("bao",), cite="theta = (h, Om, w0) with H = 100 h E (:25-28), RD fixed (:8)"),
    "bao/desi_fs_lya_union3_cc.py": Recipe(
        ("fcc", "offset", "H0", "rd", "Om", "v"), sn=dict(z_turn=0.2), bao=dict(dh_exact=True, rd="free"), cc=dict(),
        cite="theta :126-136"),
    "bao/desi_pantheon_cc.py": Recipe(
        ("H0", "offset", "rd", "Om", "v", "fcc"), sn=dict(z_turn=None, vel_mult=True), bao=dict(dh_exact=True, rd="free"), cc=dict(),
        bounds=[(40.0, 90.0), (-20.0, -19.0), (115.0, 170.0), (0.0, 1.0), (-1.3, 3.5), (0.4, 2.5)],
        cite="z_cosmo = max((1 + z)(1 + z_pec) - 1, 1e-8) (:84-90), bounds :93-102"),
    "bao/desi_des5y_obh2_theta_star.py": Recipe(
        ("offset", "H0", "obh2", "och2", "w0"), physical=True, fde="thawing", comp="PLANCK_ACT", sn=dict(z_turn=None),
        bao=dict(dh_exact=True, rd="fit"), cmb=dict(components=(1, 2), sub="inv"),
        bounds=[(-0.4, 0.4), (50.0, 90.0), (0.010, 0.030), (0.05, 0.30), (-1.0, -1 / 3)],
        cite="delta[1:] @ inv_cov_mat[1:, 1:] @ delta[1:] (:104), bounds :115-123"),
    "bao/desi_pantheon_obh2_theta_star.py": Recipe(
        ("offset", "H0", "obh2", "och2", "w0"), physical=True, fde="thawing", comp="EARLY_LCDM", sn=dict(z_turn=None),
        bao=dict(dh_exact=True, rd="fit"), cmb=dict(components=(0, 1), sub="cov"),
        bounds=[(-20.0, -19.0), (50.0, 90.0), (0.0, 0.05), (0.05, 0.30), (-1.0, -1 / 3)],
        cite="(theta*, omega_b) with the inverse of the 2 x 2 sub-covariance (:22-23), bounds :128-137"),
    "bao/desi_union3_obh2_theta_star.py": Recipe(
        _T5, physical=True, comp="PLANCK_ACT", sn=dict(z_turn=0.2), bao=dict(dh_exact=True, rd="fit"),
        cmb=dict(components=(1, 2), sub="cov"), cite="(l_A, omega_b) with inv(covariance[1:, 1:]) (:17,128)"),
    "bao/desi_des5y_omh2.py": Recipe(
        ("offset", "rd", "H0", "Om", "v"), omh2=True, sn=dict(z_turn=0.10563), bao=dict(dh_exact=True, rd="free"),
        cite="Om = omega_m / h^2 (:31-32), D_H = c / H (:41-42), step at z = 0.10563 (:88); theta :130-134"),
    "bao/desi_pantheon_rd.py": Recipe(
        ("offset", "H0", "Om", "rd", "w0"), fde="thawing", sn=dict(z_turn=None), bao=dict(dh_exact=True, rd="free"),
        bounds=[(-20.0, -19.0), (50.0, 100.0), (0.2, 0.7), (144.0, 150.0), (-1.0, -1 / 3)], gauss=[("rd", 147.14, 0.29)],
        cite="theta = (M, H0, Om, rd, w0), no velocity term (:74-76), bounds :79-87, Planck + ACT prior on r_d (:117)"),
    "bao/desi_union3_omh2.py": Recipe(
        ("offset", "rd", "H0", "Om", "v"), omh2=True, sn=dict(z_turn=0.2), bao=dict(dh_exact=True, rd="free"),
        cite="Om = omega_m / (H0 / 100)^2 (:29-31), step at z = 0.2 (:73), explicit inverse of the 22-bin covariance (:11)"),
    "bao/desi_union3_rd.py": Recipe(
        ("offset", "rd", "H0", "Om", "v"), sn=dict(z_turn=0.2), bao=dict(dh_exact=False, rd="free"),
        cite="PCHIP D_H (:48-49), step at z = 0.2 (:78)"),
    "ohd/cc_cmb.py": Recipe(
        ("H0", "obh2", "och2", "fcc"), physical=True, comp="PLANCK_ACT", cmb=dict(), cc=dict(), z_max_of=("cc",),
        bounds=[(63.0, 73.0), (0.0210, 0.0235), (0.05, 0.30), (0.30, 2.75)], prior_normalised=False,
        cite="bounds :42-49, log_prior = 0.0 inside the box (:70-73)"),
    "ohd/cc_pantheon.py": Recipe(
        ("fcc", "H0", "offset", "Om", "w0"), fde="thawing", sn=dict(z_turn=None), cc=dict(f_inverse=True), z_max_of=("sn",),
        bounds=[(0.1, 1.5), (55.0, 80.0), (-20.0, -19.0), (0.15, 0.70), (-1.0, -1 / 3)],
        cite="chi_cc * f_cc^-2 (:64), + 2 N ln f_cc (:92), bounds :69-78"),
    "ohd/cc_union3.py": Recipe(
        ("fcc", "offset", "H0", "Om", "v"), sn=dict(z_turn=0.2), scale={"v": 0.01}, cc=dict(), z_max_of=("sn",), z_pad=0.0,
        cite="v in km/s (:53), the grid ends AT max(z_cmb) (:19): the last SN takes the extrapolation branch"),
    "sn/union3_1_cmb.py": Recipe(_T5, physical=True, comp="PLANCK_ACT", sn=dict(z_turn=0.2), cmb=dict(), z_max_of=("sn",)),
}


class Joint(_Base):
    """A reference script assembled from its recipe.  ``sn`` = (z_cmb, z_hel, obs, cov) [or a dict with ``chol``];
    ``bao`` = (structured array, covariance) or (z, val, qty, inv_cov); ``cc`` = (z, H, cov)."""

    def __init__(self, recipe: Recipe, *, sn=None, bao=None, cc=None, device=0, devices=None, solve="auto", bounds=None):
        r = self.recipe = recipe
        comp = getattr(cmb_data, r.comp) if r.comp else None
        params = {name: Param(i, r.scale.get(name, 1.0)) for i, name in enumerate(r.theta)}
        params.update({name: Param(fixed=val) for name, val in r.fixed.items()})
        self.ndim = len(r.theta)
        idx = {name: i for i, name in enumerate(r.theta)}
        self.bounds = None if (bounds is None and r.bounds is None) else np.asarray(r.bounds if bounds is None else bounds, float)
        z_tops, kw = {}, {}
        if r.sn is not None:
            if isinstance(sn, dict):
                z_cmb, z_hel, obs, chol = sn["z_cmb"], sn["z_hel"], sn["obs"], sn["chol"]
            else:
                z_cmb, z_hel, obs, cov = sn
                chol = np.linalg.cholesky(np.asarray(cov, float))
            z_turn = r.sn.get("z_turn")
            kw["sn"] = dict(z_cmb=z_cmb, z_hel=z_hel, obs=obs, chol=chol, z_turn=np.inf if z_turn is None else z_turn,
                            vel_mult=r.sn.get("vel_mult", False))
            z_tops["sn"] = float(np.max(z_cmb))
        if r.bao is not None:
            bz, bv, bq, binv = bao_arrays(*bao) if len(bao) == 2 else bao
            rd = r.bao["rd"]
            kw["bao"] = dict(z=bz, val=bv, qty=bq, inv_cov=binv, dh_exact=r.bao["dh_exact"])
            if rd == "fit":
                kw["bao"]["rd_fit"] = comp["rd_fit"]
            elif rd in ("fit_late", "fit_late_plain"):
                kw["bao"].update(rd_fit=RD_FIT_PLAIN if rd == "fit_late_plain" else comp["rd_fit"], rd_wm_late=True)
            z_tops["bao"] = float(np.max(bz))
        if r.cc is not None:
            cz, ch, ccov = cc
            kw["cc"] = dict(z=cz, h=ch, inv_cov=np.linalg.inv(ccov), logdet=np.linalg.slogdet(ccov)[1],
                            f_inverse=r.cc.get("f_inverse", False))
            z_tops["cc"] = float(np.max(cz))
        if r.cmb is not None:
            comps = r.cmb.get("components")
            inv = np.asarray(comp["cmb_inv_cov"], float)
            if comps is not None:
                ii = np.ix_(list(comps), list(comps))
                sub = inv[ii] if r.cmb.get("sub", "cov") == "inv" else np.linalg.inv(np.asarray(comp["cmb_cov"])[ii])
                inv = np.zeros((3, 3))
                inv[ii] = sub  # the omitted components drop out of the 3 x 3 quadratic form exactly
            kw["cmb"] = dict(mode=comp["cmb_mode"], prior=comp["cmb_prior"], inv_cov=inv, zstar_fit=comp["zstar_fit"])
        if r.physical:
            kw["physical"] = _physical(comp)
        self.z_max = max(z_tops[b] for b in r.z_max_of) + r.z_pad
        self.engine = LikelihoodEngine(
            ndim=self.ndim, z_max=self.z_max, n_grid=N_GRID, ez_model=L.CF_EZ_PHYSICAL if r.physical else L.CF_EZ_LATE_FLAT,
            fde=FDE_BY_NAME[r.fde], params=params, bounds=self.bounds, prior_normalised=r.prior_normalised, om_mode=int(r.omh2),
            gauss=[(idx[s], m, sg) for s, m, sg in r.gauss], chi2_gauss=[(idx[s], m, sg) for s, m, sg in r.chi2_gauss],
            device=device, devices=devices, solve_mode=solve_mode_of(solve), **kw)


def build(script: str, **data) -> Joint:
    """``build("bao/desi_des5y_cc.py", sn=..., bao=..., cc=...)``."""
    if script not in RECIPES:
        raise KeyError(f"no recipe for {script!r}; known: {sorted(RECIPES)}")
    return Joint(RECIPES[script], **data)
