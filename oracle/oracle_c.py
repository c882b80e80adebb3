"""
ctypes binding of oracle/_build/libcosmofit_oracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY — see the header of cosmofit_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcosmofit_oracle.so")
NSLOTS = 10
SLOT_NAMES = ("offset", "H0", "Om", "obh2", "och2", "w0", "wa", "v", "rd", "fcc")


class _Slot(C.Structure):
    _fields_ = [("idx", C.c_int32), ("pad", C.c_int32), ("scale", C.c_double), ("fixed", C.c_double)]


class _Desc(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32), ("n_grid", C.c_int32), ("ez_model", C.c_int32), ("fde", C.c_int32),
        ("z_max", C.c_double), ("c", C.c_double),
        ("slot", _Slot * NSLOTS),
        ("n_sn", C.c_int64),
        ("z_cmb", C.c_void_p), ("z_hel", C.c_void_p), ("obs", C.c_void_p), ("step", C.c_void_p),
        ("chol", C.c_void_p), ("ld", C.c_int64),
        ("bounds", C.c_void_p),
        ("n_gauss", C.c_int32), ("pad", C.c_int32),
        ("gauss", C.c_void_p),
        ("has_vstep", C.c_int32), ("cpl_wall", C.c_int32),
        ("or_h2", C.c_double), ("omnu_h2", C.c_double), ("o_gamma_h2", C.c_double), ("nu_m0", C.c_double),
        ("nu_rho0", C.c_double), ("nu_qs_sq", C.c_double * 5), ("nu_ws", C.c_double * 5),
        ("n_bao", C.c_int32), ("bao_dh_exact", C.c_int32), ("rd_from_fit", C.c_int32), ("pad2", C.c_int32),
        ("bao_z", C.c_void_p), ("bao_val", C.c_void_p), ("bao_inv_cov", C.c_void_p), ("bao_qty", C.c_void_p),
        ("rd_fit", C.c_double * 11),
        ("cmb_mode", C.c_int32), ("n_gl", C.c_int32),
        ("gl_x", C.c_void_p), ("gl_w", C.c_void_p),
        ("cmb_prior", C.c_double * 3), ("cmb_inv_cov", C.c_double * 9), ("zstar_fit", C.c_double * 4),
        ("n_chi2_gauss", C.c_int32), ("pad3", C.c_int32),
        ("chi2_gauss", C.c_void_p),
        ("fixed_mu", C.c_void_p),
        ("n_cc", C.c_int32), ("pad4", C.c_int32),
        ("cc_z", C.c_void_p), ("cc_h", C.c_void_p), ("cc_inv_cov", C.c_void_p),
        ("cc_logdet", C.c_double),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cosmofit_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.co_solve_triangular.restype = C.c_double
        _lib.co_eval_batch.restype = C.c_int
        _lib.co_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class COracle:
    """Built from an ``oracle_np.Likelihood`` so both oracles share one description of a case."""

    def __init__(self, lk):
        if getattr(lk, "lin_coef", None) is not None or getattr(lk, "dirs", None) is not None or getattr(lk, "om_mode", 0) \
                or getattr(lk, "rd_wm_late", False) or getattr(lk, "fs8_z", None) is not None:
            raise NotImplementedError("the C oracle restates the BASELINE configs; the parameterisation variants "
                                      "(linear magnitude term, direction-dependent velocity, omega_m slot) are checked "
                                      "against oracle_np and the golden fixtures")
        self.lk = lk
        self._keep = dict(
            z_cmb=_f64(lk.z_cmb), z_hel=_f64(lk.z_hel), obs=_f64(lk.obs), step=_f64(lk.step),
            chol=_f64(lk.chol), bounds=_f64(lk.bounds),
            gauss=_f64(np.array(lk.gauss, dtype=np.float64).reshape(-1, 3)) if len(lk.gauss) else None,
        )
        k = self._keep
        d = _Desc()
        d.ndim, d.n_grid, d.ez_model, d.fde = lk.ndim, lk.n_grid, lk.ez_model, lk.fde
        d.z_max, d.c = lk.z_max, lk.c
        for i, name in enumerate(SLOT_NAMES):
            s = getattr(lk, name)
            d.slot[i].idx, d.slot[i].scale, d.slot[i].fixed = s.idx, s.scale, s.fixed
        d.n_sn = 0 if lk.z_cmb is None else len(lk.z_cmb)
        d.z_cmb, d.z_hel, d.obs, d.step = _p(k["z_cmb"]), _p(k["z_hel"]), _p(k["obs"]), _p(k["step"])
        d.chol, d.ld = _p(k["chol"]), (0 if k["chol"] is None else k["chol"].shape[1])
        d.bounds = _p(k["bounds"])
        d.n_gauss = 0 if k["gauss"] is None else len(k["gauss"])
        d.gauss = _p(k["gauss"])
        d.has_vstep, d.cpl_wall = int(lk.has_vstep), int(lk.cpl_wall)
        d.or_h2, d.omnu_h2, d.o_gamma_h2, d.nu_m0, d.nu_rho0 = lk.or_h2, lk.omnu_h2, lk.o_gamma_h2, lk.nu_m0, lk.nu_rho0
        if lk.nu_qs_sq is not None:
            d.nu_qs_sq[:] = list(lk.nu_qs_sq)
            d.nu_ws[:] = list(lk.nu_ws)
        if lk.bao_z is not None:
            k.update(bao_z=_f64(lk.bao_z), bao_val=_f64(lk.bao_val), bao_inv_cov=_f64(lk.bao_inv_cov),
                     bao_qty=np.ascontiguousarray(lk.bao_qty, dtype=np.int32))
            d.n_bao, d.bao_dh_exact = len(k["bao_z"]), int(lk.bao_dh_exact)
            d.bao_z, d.bao_val, d.bao_inv_cov, d.bao_qty = _p(k["bao_z"]), _p(k["bao_val"]), _p(k["bao_inv_cov"]), _p(k["bao_qty"])
            d.rd_from_fit = int(lk.rd_fit is not None)
            if lk.rd_fit is not None:
                d.rd_fit[:] = list(lk.rd_fit)
        d.cmb_mode = lk.cmb_mode
        if lk.cmb_mode:
            k.update(gl_x=_f64(lk.gl_x), gl_w=_f64(lk.gl_w))
            d.n_gl, d.gl_x, d.gl_w = len(k["gl_x"]), _p(k["gl_x"]), _p(k["gl_w"])
            d.cmb_prior[:] = list(lk.cmb_prior)
            d.cmb_inv_cov[:] = list(np.asarray(lk.cmb_inv_cov).ravel())
            d.zstar_fit[:] = list(lk.zstar_fit)
        k["chi2_gauss"] = _f64(np.array(lk.chi2_gauss, dtype=np.float64).reshape(-1, 3)) if len(lk.chi2_gauss) else None
        d.n_chi2_gauss = 0 if k["chi2_gauss"] is None else len(k["chi2_gauss"])
        d.chi2_gauss = _p(k["chi2_gauss"])
        k["fixed_mu"] = _f64(lk.fixed_mu)
        d.fixed_mu = _p(k["fixed_mu"])
        if lk.cc_z is not None:
            k.update(cc_z=_f64(lk.cc_z), cc_h=_f64(lk.cc_h), cc_inv_cov=_f64(lk.cc_inv_cov))
            d.n_cc, d.cc_z, d.cc_h, d.cc_inv_cov = len(k["cc_z"]), _p(k["cc_z"]), _p(k["cc_h"]), _p(k["cc_inv_cov"])
            d.cc_logdet = float(lk.cc_logdet)
        self.d = d
        self.threads_used = 1

    def eval(self, thetas, out_kind=0, nthreads=0):
        th = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        out = np.empty(len(th))
        self.threads_used = lib().co_eval_batch(C.byref(self.d), _p(th), C.c_int64(len(th)), _p(out),
                                                C.c_int(out_kind), C.c_int(nthreads))
        return out

    def chi2(self, thetas, nthreads=0):
        return self.eval(thetas, 0, nthreads)

    def logp(self, thetas, nthreads=0):
        return self.eval(thetas, 2, nthreads)

    def logl(self, thetas, nthreads=0):
        return self.eval(thetas, 1, nthreads)

    def blocks(self, theta):
        """(chi2 blocks [sn, bao, cmb], bao theory vector, cmb distance vector) of one walker."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        b, bt, cv = np.zeros(3), np.zeros(max(self.d.n_bao, 1)), np.zeros(3)
        lib().co_blocks(C.byref(self.d), _p(th), _p(b), _p(bt), _p(cv))
        return b, bt[: self.d.n_bao], cv

    def sn_parts(self, theta):
        n, G = self.d.n_sn, self.d.n_grid
        th = np.ascontiguousarray(theta, dtype=np.float64)
        dm, mc, dl, cum, dh = np.empty(n), np.empty(n), np.empty(n), np.empty(G), np.empty(G)
        lib().co_sn_parts(C.byref(self.d), _p(th), _p(dm), _p(mc), _p(dl), _p(cum), _p(dh))
        return dict(dm=dm, mu_corr=mc, delta=dl, cum=cum, dh=dh)


def interp_hermite(xq, x, y, yp):
    xq, x, y, yp = map(_f64, (xq, x, y, yp))
    out = np.empty_like(xq)
    lib().co_interp_hermite(_p(xq), C.c_int64(xq.size), _p(x), _p(y), _p(yp), C.c_int64(x.size), _p(out))
    return out


def interp_pchip(xq, x, y):
    xq, x, y = map(_f64, (xq, x, y))
    out = np.empty_like(xq)
    lib().co_interp_pchip(_p(xq), C.c_int64(xq.size), _p(x), _p(y), C.c_int64(x.size), _p(out))
    return out


def pchip_slopes(x, y):
    x, y = map(_f64, (x, y))
    d = np.empty_like(x)
    lib().co_pchip_slopes(_p(x), _p(y), C.c_int64(x.size), _p(d))
    return d


def solve_triangular(L, b):
    L, b = _f64(L), _f64(b)
    y = np.empty_like(b)
    return lib().co_solve_triangular(_p(L), C.c_int64(b.size), C.c_int64(L.shape[1]), _p(b), _p(y))


def max_threads() -> int:
    return lib().co_max_threads()
