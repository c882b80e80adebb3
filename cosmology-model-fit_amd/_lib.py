"""
ctypes binding of the gfx950 shared library (csrc/ -> libcosmofit_hip.so) behind include/cosmofit.h.

There is no CPU implementation behind this module: if the library is missing or no MI355X is
visible, every operation raises (``CosmofitError``), it never falls back.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# COSMOFIT_LIB: another build of the same sources (tools/build_variant.sh: tuning / debug variants for A/B timing)
LIB_PATH = os.environ.get("COSMOFIT_LIB") or os.path.join(_HERE, "libcosmofit_hip.so")
CSRC = os.path.join(_HERE, "csrc")

CF_ABI_VERSION = 8
CF_P_NSLOTS = 15
SLOTS = ("offset", "H0", "Om", "obh2", "och2", "w0", "wa", "v", "rd", "fcc", "lin", "v2", "v3", "s8", "fs8err")

# enums of include/cosmofit.h
CF_EZ_LATE_FLAT, CF_EZ_PHYSICAL = 0, 1
CF_FDE_LCDM, CF_FDE_WCDM, CF_FDE_THAWING, CF_FDE_CPL = 0, 1, 2, 3
CF_OUT_CHI2, CF_OUT_LOGL, CF_OUT_LOGP = 0, 1, 2
CF_SOLVE_BLOCKED_TRSM, CF_SOLVE_INVERSE_GEMM, CF_SOLVE_AUTO = 0, 1, 2
SOLVE_MODES = {"blocked": CF_SOLVE_BLOCKED_TRSM, "inverse": CF_SOLVE_INVERSE_GEMM, "auto": CF_SOLVE_AUTO}
CF_CMB_NONE = 0
STATUS = {0: "CF_OK", -1: "CF_ERR_INVALID", -2: "CF_ERR_NO_DEVICE", -3: "CF_ERR_HIP", -4: "CF_ERR_NOT_POSDEF",
          -5: "CF_ERR_UNSUPPORTED", -6: "CF_ERR_ILL_CONDITIONED"}


class CosmofitError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class cf_param(C.Structure):
    _fields_ = [("idx", C.c_int32), ("_pad", C.c_int32), ("scale", C.c_double), ("fixed", C.c_double)]


class cf_gauss_prior(C.Structure):
    _fields_ = [("idx", C.c_int32), ("_pad", C.c_int32), ("mean", C.c_double), ("sigma", C.c_double)]


class cf_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("struct_size", C.c_int32), ("device", C.c_int32), ("ndim", C.c_int32),
        ("ez_model", C.c_int32), ("fde", C.c_int32), ("n_grid", C.c_int32), ("_pad0", C.c_int32),
        ("z_max", C.c_double), ("c_km_s", C.c_double),
        ("param", cf_param * CF_P_NSLOTS),
        ("n_sn", C.c_int64),
        ("sn_z_cmb", C.c_void_p), ("sn_z_hel", C.c_void_p), ("sn_obs", C.c_void_p), ("sn_step", C.c_void_p),
        ("sn_z_turn", C.c_double),
        ("sn_chol", C.c_void_p), ("sn_chol_ld", C.c_int64),
        ("n_bao", C.c_int32), ("bao_dh_mode", C.c_int32), ("rd_mode", C.c_int32), ("_pad1", C.c_int32),
        ("bao_z", C.c_void_p), ("bao_val", C.c_void_p), ("bao_qty", C.c_void_p), ("bao_inv_cov", C.c_void_p),
        ("rd_fit", C.c_double * 11),
        ("cmb_mode", C.c_int32), ("n_gl", C.c_int32),
        ("gl_x", C.c_void_p), ("gl_w", C.c_void_p),
        ("cmb_prior", C.c_double * 3), ("cmb_inv_cov", C.c_double * 9),
        ("zstar_fit", C.c_double * 11), ("o_gamma_h2", C.c_double),
        ("or_h2", C.c_double), ("omnu_h2", C.c_double), ("nu_m0", C.c_double), ("nu_rho0", C.c_double),
        ("nu_qs_sq", C.c_double * 5), ("nu_ws", C.c_double * 5),
        ("bounds", C.c_void_p),
        ("n_gauss", C.c_int32), ("cpl_wall", C.c_int32),
        ("gauss", C.c_void_p),
        ("n_chi2_gauss", C.c_int32), ("_pad2", C.c_int32),
        ("chi2_gauss", C.c_void_p),
        ("sn_fixed_mu", C.c_void_p),
        ("n_cc", C.c_int32), ("_pad3", C.c_int32),
        ("cc_z", C.c_void_p), ("cc_h", C.c_void_p), ("cc_inv_cov", C.c_void_p),
        ("cc_logdet", C.c_double),
        ("solve_mode", C.c_int32), ("_pad4", C.c_int32),
        ("probe_limit", C.c_double),
        ("n_devices", C.c_int32), ("_pad5", C.c_int32), ("devices", C.c_void_p),
        ("om_mode", C.c_int32), ("rd_wm_mode", C.c_int32), ("sn_lin_coef", C.c_void_p), ("sn_dir", C.c_void_p),
        ("n_fs8", C.c_int32), ("fs8_steps", C.c_int32),
        ("fs8_z", C.c_void_p), ("fs8_val", C.c_void_p), ("fs8_inv_cov", C.c_void_p), ("fs8_fid", C.c_void_p),
        ("logl_const", C.c_double), ("fs8_a_init", C.c_double),
        ("sn_vel_mode", C.c_int32), ("cc_f_mode", C.c_int32), ("prior_norm_mode", C.c_int32), ("fs8_n_agrid", C.c_int32),
    ]


class cf_info(C.Structure):
    _fields_ = [
        ("n_sn", C.c_int64), ("n_sn_pad", C.c_int64), ("packed_chol_bytes", C.c_int64),
        ("workspace_bytes", C.c_int64), ("max_walkers", C.c_int64), ("nonfinite_count", C.c_int64),
        ("device", C.c_int32), ("cu_count", C.c_int32), ("gcn_arch", C.c_char * 64),
        ("pack_probe_rel", C.c_double), ("solve_mode", C.c_int32), ("n_devices", C.c_int32),
        ("devices", C.c_int32 * 16),
    ]


# every symbol include/cosmofit.h declares: name -> (restype, argtypes)
_VP, _I64, _I32 = C.c_void_p, C.c_int64, C.c_int32
EXPORTS = {
    "cf_device_count": (C.c_int, []),
    "cf_last_error": (C.c_char_p, []),
    "cf_abi_version": (C.c_int, []),
    "cf_create": (C.c_int, [C.POINTER(cf_desc), C.POINTER(_VP)]),
    "cf_destroy": (None, [_VP]),
    "cf_get_info": (C.c_int, [_VP, C.POINTER(cf_info)]),
    "cf_eval": (C.c_int, [_VP, _VP, _I64, _VP, _I32]),
    "cf_split_rows": (None, [_I64, _I32, _I32, C.POINTER(_I64), C.POINTER(_I64)]),
    "cf_eval_device": (C.c_int, [_VP, _VP, _I64, _VP, _I32, _VP]),
    "cf_eval_parts": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "cf_eval_table": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "cf_eval_bao_at": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _VP]),
    "cf_eval_hz": (C.c_int, [_VP, _VP, _VP, _I64, _VP]),
    "cf_eval_fs8_at": (C.c_int, [_VP, _VP, _VP, _I64, _VP]),
    "cf_last_kernel_ms": (C.c_int, [_VP, C.POINTER(C.c_float * 2)]),
    "cf_enable_timing": (C.c_int, [_VP, C.c_int]),
    "cf_set_timing_stride": (C.c_int, [_VP, C.c_int]),
    "cf_timed_calls": (C.c_int64, [_VP]),
    "cf_kernel_ms": (C.c_int, [_VP, _I64, C.POINTER(C.c_float * 2)]),
    "cf_kernel_ms3": (C.c_int, [_VP, _I64, C.POINTER(C.c_float * 3)]),
    "cf_interp_hermite": (C.c_int, [_VP, _I64, _VP, _VP, _VP, _I64, _VP]),
    "cf_interp_pchip": (C.c_int, [_VP, _I64, _VP, _VP, _I64, _VP]),
    "cf_solve_triangular": (C.c_int, [_VP, _I64, _I64, _VP, _I64, _VP]),
    "cf_selftest_invpack_host": (C.c_int, [_VP, _I64, _I64, _VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cf_selftest_log10": (C.c_int, [_VP, _I64, _VP]),
    "cf_selftest_log10_tab": (C.c_int, [_VP, _I64, _VP]),
    "cf_selftest_exp_tab": (C.c_int, [_VP, _I64, _VP]),
    "cf_selftest_pos_ops": (C.c_int, [_VP, _VP, _I64, _VP]),
    "cf_ens_active_count": (_I64, [C.c_uint64, _I32, _I32, _I64, _I64]),
    "cf_ens_comp_count": (_I64, [C.c_uint64, _I32, _I32, _I64]),
    "cf_ens_active_set": (C.c_int, [C.c_uint64, _I32, _I32, _I64, _I64, _VP, _VP, _VP]),
    "cf_ens_kde_prepare": (C.c_int, [_VP, _I64, _I32, _I32, _I32, C.c_uint64, _VP, _VP, _VP]),
    "cf_ens_propose": (C.c_int, [_I32, _VP, _I64, _I32, _I32, _I32, C.c_uint64, _VP, _I64, C.c_uint64, C.c_double,
                                 C.c_double, _VP, _VP, _VP, _VP, _VP]),
    "cf_ens_accept": (C.c_int, [_VP, _VP, _I64, C.c_int32, C.c_uint64, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "cf_selftest_pack_host": (C.c_int, [_VP, _I64, _I64, _VP, C.POINTER(C.c_double), C.POINTER(_I64)]),
}


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libcosmofit_hip.so (in-tree)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (same
    SONAME as /opt/rocm's); if both copies get loaded the second one sees no GPU.  bench.py and the
    tests use torch for device tensors and torch.distributed, so when torch is installed we bind to
    ITS runtime (whichever of the two is imported first); without torch the RUNPATH (/opt/rocm) applies."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load the shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CosmofitError(-2, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(hipcc --offload-arch=gfx950); there is no CPU implementation to fall back to")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.cf_abi_version() != CF_ABI_VERSION:
            raise CosmofitError(-1, "libcosmofit_hip.so ABI version mismatch; rebuild it")
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise CosmofitError(rc, lib().cf_last_error().decode(errors="replace"))
