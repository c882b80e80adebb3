"""
Batched log-likelihood engine: the host-side object behind the emcee / nautilus callbacks.

One ``LikelihoodEngine`` = one ``cf_handle`` of the C-ABI (include/cosmofit.h): the data vectors
and the packed Cholesky factor live on one MI355X, and ``chi_squared / log_likelihood /
log_probability`` evaluate W walkers per call through the HIP kernels.

Reference call sites this replaces: ``chi_squared`` / ``log_probability`` of sn/pantheon.py:57-97
and the batch wrappers bao/desi.py:100-106, bao/desi_cmb.py:137-143.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib as L

C_KM_S = 299792.458  # scipy.constants.c / 1000 (sn/pantheon.py:12)


@dataclass
class Param:
    """A physical parameter: read from theta[idx]*scale, or fixed."""
    idx: int = -1
    scale: float = 1.0
    fixed: float = 0.0


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve_mode_of(solve="auto", latency_mode=None) -> int:
    """solve: "auto" | "inverse" | "blocked"; latency_mode (older keyword): True = "inverse", False = "blocked"."""
    if latency_mode is not None:
        solve = "inverse" if latency_mode else "blocked"
    if solve not in L.SOLVE_MODES:
        raise ValueError(f"solve must be one of {sorted(L.SOLVE_MODES)}")
    return L.SOLVE_MODES[solve]


class LikelihoodEngine:
    def __init__(self, *, ndim: int, z_max: float, n_grid: int = 4000, fde: int = L.CF_FDE_LCDM,
                 ez_model: int = L.CF_EZ_LATE_FLAT, params: dict, sn: Optional[dict] = None,
                 bao: Optional[dict] = None, cmb: Optional[dict] = None, physical: Optional[dict] = None,
                 cc: Optional[dict] = None, fs8: Optional[dict] = None, solve_mode: int = L.CF_SOLVE_AUTO,
                 bounds=None, gauss: Sequence = (), chi2_gauss: Sequence = (), cpl_wall: bool = False,
                 device: int = 0, devices=None, probe_limit: float = 0.0, om_mode: int = 0, logl_const: float = 0.0, prior_normalised: bool = True, c_km_s: float = C_KM_S):
        """
        params: {"H0": Param(1), "Om": Param(2), ...} for the slots of include/cosmofit.h (cf_param_slot).
        sn: dict(z_cmb, z_hel, obs, chol[, step | z_turn, fixed_mu, lin_coef, dirs]) — chol is
            cho_factor(cov, lower=True)[0]; the strict upper triangle is never read; fixed_mu[i] (NaN = none) replaces
            mu_theory for SN i (SH0ES Cepheid hosts, sn/pantheon_and_sh0es.py:63-69); lin_coef[i] multiplies the "lin"
            slot and is added to the offset (bulk-flow magnitude term, bao/desi_cmb_pantheon_H0trgb.py:102-106); dirs
            [N, 3] unit vectors make the peculiar velocity n . (v, v2, v3) x step (sn/pantheon_dipole_xyz.py:50-60).
        logl_const: constant added to log L (Gaussian normalisations a script keeps in its log-likelihood).
        om_mode: 1 = the "Om" slot holds omega_m = Omega_m h^2 (bao/desi_omh2.py:18-20).
        fs8: dict(z, val, inv_cov, fid, a_init[, steps, a_grid]) — growth-rate block with the slots "s8" and "fs8err"
            (fs8/fs8.py:64-125); fid[k] = H_fid(z_k) D_M,fid(z_k) of the Alcock-Paczynski correction; a_grid = N: delta' at the
            data by interp_pchip on the scripts' np.logspace(log10 a_init, 0, N) grid (fs8/fs8.py:79-98), 0: read directly.
        cc: dict(z, h, inv_cov, logdet) — cosmic chronometers with the rescale parameter slot "fcc"
            (bao/desi_union3_cc_theta_star.py:129-139).
        bao: dict(z, val, qty (0 DV/rd, 1 DM/rd, 2 DH/rd, 3 F_AP), inv_cov[, dh_exact=False, rd_fit=None]);
            rd_fit = (b, m, a1..a9) selects the r_drag fitting formula, otherwise the "rd" slot is used; rd_wm_late=True hands
            it wm = Omega_m h^2 of the late-time flat model instead of omega_b + omega_c + omega_nu (bao/desi_bbn.py:46-60).
        cmb: dict(mode (1 R-lA-wb, 2 lA only, 3 theta*-wb-wm), prior[3], inv_cov[3,3], zstar_fit (s1,s2,b,m)[, n_gl=100]).
        physical: dict(or_h2, omnu_h2, o_gamma_h2, nu_m0, nu_rho0, nu_qs_sq[5], nu_ws[5]) — required by
            ez_model=CF_EZ_PHYSICAL (see cmb_data.PLANCK_ACT / EARLY_LCDM).
        gauss / chi2_gauss: sequences of (idx, mean, sigma).
        devices: None (the single ordinal `device`), "all" (every visible GPU) or a sequence of HIP ordinals: the data and
            the packed factor are replicated on each, and the host-buffer calls (chi_squared / log_likelihood /
            log_probability) split their rows over them -- one host process, as emcee / nautilus run, on every GPU.
        probe_limit: acceptance limit of the create-time accuracy probe of the explicit inverse (0 = default 1e-11).
        solve_mode: CF_SOLVE_AUTO (default: the inverse-GEMM solve when its create-time probe passes, otherwise the
            blocked forward substitution), CF_SOLVE_INVERSE_GEMM or CF_SOLVE_BLOCKED_TRSM; info()["solve_mode"]
            tells which one runs.  See solve_mode_of() for the keyword form the mirrors take.
        """
        lib = L.lib()
        d = L.cf_desc()
        d.abi_version, d.struct_size = L.CF_ABI_VERSION, C.sizeof(L.cf_desc)
        d.device, d.ndim = device, ndim
        d.ez_model, d.fde, d.n_grid = ez_model, fde, n_grid
        d.z_max, d.c_km_s = float(z_max), float(c_km_s)
        d.solve_mode = int(solve_mode)
        d.probe_limit = float(probe_limit)
        d.om_mode = int(om_mode)
        d.logl_const = float(logl_const)
        d.prior_norm_mode = 0 if prior_normalised else 1
        dev_arr = None
        if devices is not None:
            if isinstance(devices, str):
                if devices != "all":
                    raise ValueError('devices must be None, "all" or a sequence of device ordinals')
                d.n_devices = -1
            else:
                dev_arr = np.ascontiguousarray(list(devices), dtype=np.int32)
                if dev_arr.size < 1:
                    raise ValueError("devices must name at least one device")
                d.n_devices, d.devices = dev_arr.size, _ptr(dev_arr)
        unknown = set(params) - set(L.SLOTS)
        if unknown:
            raise ValueError(f"unknown parameter slots {sorted(unknown)}; valid: {L.SLOTS}")
        for i, name in enumerate(L.SLOTS):
            p = params.get(name, Param(fixed=-1.0) if name == "w0" else (Param(fixed=1.0) if name in ("fcc", "fs8err") else Param()))
            d.param[i].idx, d.param[i].scale, d.param[i].fixed = p.idx, p.scale, p.fixed
        keep = []
        if sn is not None:
            z_cmb, z_hel, obs = _f64(sn["z_cmb"]), _f64(sn["z_hel"]), _f64(sn["obs"])
            chol = _f64(sn["chol"])
            if chol.ndim != 2 or chol.shape[0] != chol.shape[1] or chol.shape[0] != z_cmb.size:
                raise ValueError("sn['chol'] must be (N, N) with N = len(z_cmb)")
            step = None if sn.get("step") is None else _f64(sn["step"])
            keep += [z_cmb, z_hel, obs, chol, step]
            d.n_sn = z_cmb.size
            d.sn_z_cmb, d.sn_z_hel, d.sn_obs, d.sn_step = _ptr(z_cmb), _ptr(z_hel), _ptr(obs), _ptr(step)
            d.sn_z_turn = float(sn.get("z_turn", 0.15))
            d.sn_vel_mode = 1 if sn.get("vel_mult", False) else 0
            d.sn_chol, d.sn_chol_ld = _ptr(chol), chol.shape[1]
            if sn.get("fixed_mu") is not None:
                fm = _f64(sn["fixed_mu"])
                if fm.size != z_cmb.size:
                    raise ValueError("sn['fixed_mu'] must have one entry per SN (NaN where unused)")
                keep.append(fm)
                d.sn_fixed_mu = _ptr(fm)
            if sn.get("lin_coef") is not None:
                lc = _f64(sn["lin_coef"])
                if lc.size != z_cmb.size:
                    raise ValueError("sn['lin_coef'] must have one entry per SN")
                keep.append(lc)
                d.sn_lin_coef = _ptr(lc)
            if sn.get("dirs") is not None:
                dirs = _f64(sn["dirs"])
                if dirs.shape != (z_cmb.size, 3):
                    raise ValueError("sn['dirs'] must be (N, 3)")
                keep.append(dirs)
                d.sn_dir = _ptr(dirs)
        self.n_bao = 0
        if physical is not None:
            d.or_h2, d.omnu_h2, d.o_gamma_h2 = physical["or_h2"], physical["omnu_h2"], physical["o_gamma_h2"]
            d.nu_m0, d.nu_rho0 = physical["nu_m0"], physical["nu_rho0"]
            d.nu_qs_sq[:] = [float(x) for x in physical["nu_qs_sq"]]
            d.nu_ws[:] = [float(x) for x in physical["nu_ws"]]
        if bao is not None:
            bz, bv, binv = _f64(bao["z"]), _f64(bao["val"]), _f64(bao["inv_cov"])
            bq = np.ascontiguousarray(bao["qty"], dtype=np.int32)
            if not (bz.size == bv.size == bq.size) or binv.shape != (bz.size, bz.size):
                raise ValueError("bao: z, val, qty must have n entries and inv_cov must be (n, n)")
            keep += [bz, bv, binv, bq]
            d.n_bao, d.bao_z, d.bao_val, d.bao_qty, d.bao_inv_cov = bz.size, _ptr(bz), _ptr(bv), _ptr(bq), _ptr(binv)
            d.bao_dh_mode = 1 if bao.get("dh_exact", False) else 0
            if bao.get("rd_fit") is not None:
                d.rd_mode = 1
                d.rd_fit[:] = [float(x) for x in bao["rd_fit"]]
                d.rd_wm_mode = 1 if bao.get("rd_wm_late", False) else 0
            self.n_bao = int(bz.size)
        if cc is not None:
            cz, ch, cinv = _f64(cc["z"]), _f64(cc["h"]), _f64(cc["inv_cov"])
            if cz.size != ch.size or cinv.shape != (cz.size, cz.size):
                raise ValueError("cc: z, h must have n entries and inv_cov must be (n, n)")
            keep += [cz, ch, cinv]
            d.n_cc, d.cc_z, d.cc_h, d.cc_inv_cov, d.cc_logdet = cz.size, _ptr(cz), _ptr(ch), _ptr(cinv), float(cc["logdet"])
            d.cc_f_mode = 1 if cc.get("f_inverse", False) else 0
        self.n_fs8 = 0
        if fs8 is not None:
            fz, fv, finv, ffid = _f64(fs8["z"]), _f64(fs8["val"]), _f64(fs8["inv_cov"]), _f64(fs8["fid"])
            if not (fz.size == fv.size == ffid.size) or finv.shape != (fz.size, fz.size):
                raise ValueError("fs8: z, val, fid must have n entries and inv_cov must be (n, n)")
            keep += [fz, fv, finv, ffid]
            d.n_fs8, d.fs8_z, d.fs8_val, d.fs8_inv_cov, d.fs8_fid = fz.size, _ptr(fz), _ptr(fv), _ptr(finv), _ptr(ffid)
            d.fs8_a_init, d.fs8_steps = float(fs8["a_init"]), int(fs8.get("steps", 0))
            d.fs8_n_agrid = int(fs8.get("a_grid", 0))
            self.n_fs8 = int(fz.size)
        if cmb is not None:
            from .cmb_data import ZSTAR_CONSTS
            gx, gw = np.polynomial.legendre.leggauss(int(cmb.get("n_gl", 100)))  # cmb/data_planck_act_compression.py:150
            gx, gw = _f64(gx), _f64(gw)
            keep += [gx, gw]
            d.cmb_mode, d.n_gl, d.gl_x, d.gl_w = int(cmb["mode"]), gx.size, _ptr(gx), _ptr(gw)
            d.cmb_prior[:] = [float(x) for x in cmb["prior"]]
            d.cmb_inv_cov[:] = [float(x) for x in np.asarray(cmb["inv_cov"], dtype=np.float64).ravel()]
            d.zstar_fit[:] = [float(x) for x in tuple(cmb["zstar_fit"]) + ZSTAR_CONSTS]
        self.bounds = None if bounds is None else _f64(bounds).reshape(ndim, 2)
        d.bounds = _ptr(self.bounds)
        g = (L.cf_gauss_prior * max(len(gauss), 1))()
        for k, (idx, mean, sigma) in enumerate(gauss):
            g[k].idx, g[k].mean, g[k].sigma = int(idx), float(mean), float(sigma)
        d.n_gauss, d.gauss = len(gauss), C.cast(g, C.c_void_p)
        cg = (L.cf_gauss_prior * max(len(chi2_gauss), 1))()
        for k, (idx, mean, sigma) in enumerate(chi2_gauss):
            cg[k].idx, cg[k].mean, cg[k].sigma = int(idx), float(mean), float(sigma)
        d.n_chi2_gauss, d.chi2_gauss = len(chi2_gauss), C.cast(cg, C.c_void_p)
        d.cpl_wall = int(cpl_wall)
        self.ndim = ndim
        self.n_grid, self.z_max = int(n_grid), float(z_max)
        self.n_sn = int(d.n_sn)
        self._h = C.c_void_p()
        self._cf_eval = lib.cf_eval
        L.check(lib.cf_create(C.byref(d), C.byref(self._h)))
        del keep  # cf_create copied everything

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib().cf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- evaluation -------------------------------------------------------------------------
    def _eval(self, theta, kind):
        # the sampler's callback: a 16-walker call is ~40 us of which the wrapper should not be five (no atleast_2d, raw
        # addresses instead of ctypes pointer objects: 6 -> 3 us per call)
        th = theta if (type(theta) is np.ndarray and theta.dtype == np.float64 and theta.flags.c_contiguous) \
            else np.ascontiguousarray(theta, dtype=np.float64)
        single = th.ndim == 1
        if th.ndim not in (1, 2) or th.shape[-1] != self.ndim:
            raise ValueError(f"theta must have {self.ndim} columns, got {th.shape}")
        W = 1 if single else th.shape[0]
        out = np.empty(W, dtype=np.float64)
        rc = self._cf_eval(self._h, th.ctypes.data, W, out.ctypes.data, kind)
        if rc:
            L.check(rc)
        return float(out[0]) if single else out

    def chi_squared(self, theta):
        """chi^2(theta) for one theta [ndim] -> float, or a batch [W, ndim] -> float64[W]."""
        return self._eval(theta, L.CF_OUT_CHI2)

    def log_likelihood(self, theta):
        """-0.5 chi^2 (what nautilus wants: prior handled by the sampler)."""
        return self._eval(theta, L.CF_OUT_LOGL)

    def log_probability(self, theta):
        """log prior + log L; -inf outside the strict box (emcee callback, vectorize=True capable)."""
        return self._eval(theta, L.CF_OUT_LOGP)

    def eval_device(self, theta_ptr: int, W: int, out_ptr: int, kind: int = L.CF_OUT_LOGP, stream: int = 0):
        """Asynchronous evaluation on device-resident buffers (raw device pointers, e.g. tensor.data_ptr())."""
        L.check(L.lib().cf_eval_device(self._h, C.c_void_p(theta_ptr), W, C.c_void_p(out_ptr), kind,
                                       C.c_void_p(stream)))

    def torch_log_prob(self, kind: int = L.CF_OUT_LOGP):
        """Callable ``f(theta: cuda float64 tensor [W, ndim]) -> tensor [W]`` evaluating on the tensor's device
        with the HIP kernels, asynchronously on torch's current stream (for ``ensemble.ShardedEnsemble``)."""
        import torch

        def f(theta):
            if not theta.is_cuda or theta.dtype != torch.float64 or not theta.is_contiguous():
                raise ValueError("theta must be a contiguous float64 tensor on the engine's GPU")
            out = torch.empty(theta.shape[0], dtype=torch.float64, device=theta.device)
            if theta.shape[0]:
                self.eval_device(theta.data_ptr(), theta.shape[0], out.data_ptr(), kind,
                                 torch.cuda.current_stream(theta.device).cuda_stream)
            return out

        return f

    def parts(self, theta):
        """Intermediates of a (small) batch — for plots and tests: DM(z_cmb), mu_corr, residual (SN block);
        chi2_blocks[:, 0:3] = (sn, bao, cmb), cmb_vector = the compressed-CMB theory 3-vector, chi2_cc, chi2_fs8, bao_theory,
        fs8_theory (before the Alcock-Paczynski division)."""
        th = np.atleast_2d(_f64(theta))
        W, n, nb, nf = th.shape[0], self.n_sn, self.n_bao, self.n_fs8
        dm = np.empty((W, n)) if n else None
        mc = np.empty((W, n)) if n else None
        dl = np.empty((W, n)) if n else None
        bt = np.empty((W, nb)) if nb else None
        ft = np.empty((W, nf)) if nf else None
        blocks = np.empty((W, 10))
        L.check(L.lib().cf_eval_parts(self._h, _ptr(th), W, _ptr(dm), _ptr(mc), _ptr(dl), _ptr(blocks), _ptr(bt), _ptr(ft)))
        return dict(dm=dm, mu_corr=mc, delta=dl, chi2_blocks=blocks[:, :3], cmb_vector=blocks[:, 3:6], chi2_cc=blocks[:, 6],
                    chi2_fs8=blocks[:, 7], z_star=blocks[:, 8], r_drag=blocks[:, 9], bao_theory=bt, fs8_theory=ft)

    def distance_table(self, theta):
        """(z_grid [G], cum_dm [W, G], dh_grid [W, G]) of one theta or a small batch: the grid of the scripts
        (``np.linspace(0, z_max, 4000)``) and the two arrays their ``DM_z`` interpolates (sn/pantheon.py:34-40)."""
        th = np.atleast_2d(_f64(theta))
        W, G = th.shape[0], self.n_grid
        cum, dh = np.empty((W, G)), np.empty((W, G))
        L.check(L.lib().cf_eval_table(self._h, _ptr(th), W, _ptr(cum), _ptr(dh)))
        return np.linspace(0, self.z_max, num=G), cum, dh

    def DM_z(self, theta, z):
        """Comoving distance at arbitrary redshifts: the reference's ``DM_z(params, z)`` (sn/pantheon.py:34-40) -- the
        walker's table from the GPU, then the stand-alone GPU Hermite operator (interpolator.py:117-119)."""
        from .interpolator import interp_hermite

        z_grid, cum, dh = self.distance_table(np.asarray(theta, dtype=np.float64).reshape(1, -1))
        return interp_hermite(np.atleast_1d(_f64(z)), z_grid, cum[0], dh[0])

    def bao_theory_at(self, theta, z, qty):
        """``bao_theory(z, qty, params)`` of the scripts at arbitrary redshifts for one theta (bao/desi.py:38-56): the smooth curves
        of the post-fit plots (bao/plot_predictions.py:24-45).  qty: 0 D_V / r_d, 1 D_M / r_d, 2 D_H / r_d, 3 F_AP."""
        th = _f64(theta).reshape(-1)
        if th.size != self.ndim:
            raise ValueError(f"theta must have {self.ndim} entries")
        z = np.atleast_1d(_f64(z))
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(qty), z.shape), dtype=np.int32)
        out = np.empty(z.size)
        L.check(L.lib().cf_eval_bao_at(self._h, _ptr(th), _ptr(z), _ptr(q), z.size, _ptr(out)))
        return out

    def H_z(self, theta, z):
        """``H_z(z, params)`` of the scripts in km/s/Mpc at arbitrary redshifts for one theta (ohd/cc.py:95-96)."""
        th = _f64(theta).reshape(-1)
        if th.size != self.ndim:
            raise ValueError(f"theta must have {self.ndim} entries")
        z = np.atleast_1d(_f64(z))
        out = np.empty(z.size)
        L.check(L.lib().cf_eval_hz(self._h, _ptr(th), _ptr(z), z.size, _ptr(out)))
        return out

    def fs8_theory_at(self, theta, z):
        """f sigma_8 (before the Alcock-Paczynski division) at arbitrary redshifts for one theta: ``fs8_theory(1 / (1 + z), params)``
        of the growth-rate scripts (fs8/fs8.py:84-98,221-226)."""
        th = _f64(theta).reshape(-1)
        if th.size != self.ndim:
            raise ValueError(f"theta must have {self.ndim} entries")
        z = np.atleast_1d(_f64(z))
        out = np.empty(z.size)
        L.check(L.lib().cf_eval_fs8_at(self._h, _ptr(th), _ptr(z), z.size, _ptr(out)))
        return out

    def enable_timing(self, slots=1, stride=1):
        """Keep HIP-event timings of the last `slots` timed evaluations (0 = off); only every `stride`-th evaluation is timed."""
        L.check(L.lib().cf_enable_timing(self._h, int(slots)))
        L.check(L.lib().cf_set_timing_stride(self._h, int(stride)))

    def kernel_ms(self):
        """[(residual_ms, solve_ms), ...] for every evaluation still in the timing ring."""
        lib = L.lib()
        n = lib.cf_timed_calls(self._h)
        out = []
        t = (C.c_float * 2)()
        for call in range(n):
            if lib.cf_kernel_ms(self._h, call, C.byref(t)) == 0:
                out.append((float(t[0]), float(t[1])))
        return out

    def kernel_ms3(self):
        """[(walker_kernel_ms, small_blocks_ms, solve_ms), ...] for every evaluation still in the timing ring."""
        lib = L.lib()
        out = []
        t = (C.c_float * 3)()
        for call in range(lib.cf_timed_calls(self._h)):
            if lib.cf_kernel_ms3(self._h, call, C.byref(t)) == 0:
                out.append((float(t[0]), float(t[1]), float(t[2])))
        return out

    def last_kernel_ms(self):
        t = (C.c_float * 2)()
        L.check(L.lib().cf_last_kernel_ms(self._h, C.byref(t)))
        return float(t[0]), float(t[1])

    def info(self) -> dict:
        i = L.cf_info()
        L.check(L.lib().cf_get_info(self._h, C.byref(i)))
        out = {k: (getattr(i, k).decode() if k == "gcn_arch" else getattr(i, k)) for k, _ in L.cf_info._fields_}
        out["devices"] = list(i.devices)[: min(int(i.n_devices), 16)]
        return out
