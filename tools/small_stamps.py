"""Per-workgroup timeline of the small-batch solve kernel from in-kernel s_memtime stamps (debug build, -DCF_TRSM_STAMPS).
usage (GPU box): tools/build_variant.sh stamps -DCF_TRSM_STAMPS && COSMOFIT_LIB=.../libcosmofit_hip_stamps.so python tools/small_stamps.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("cosmology-model-fit_amd")
syn = amd.synthetic
data = syn.pantheon_like(int(os.environ.get("N_SN", 1701)), seed=0)
lk = amd.sn_pantheon.PantheonLikelihood(data["z_cmb"], data["z_hel"], data["obs"], chol=data["chol"])
th = syn.walkers(lk.bounds, 4096, seed=1)
lk.log_probability(th)
for _ in range(200):
    lk.log_probability(th[:16])
buf = (C.c_uint64 * (128 * 4 * 8))()
assert amd._lib.lib().cf_debug_small_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(128, 4, 8).astype(np.int64)
n_units = int((st[:, 0, 0] > 0).sum())
t0 = st[:n_units, :, 0].min()
w0 = st[:n_units, :, 6].min()
print("unit rb j | start (rel) | loads + MFMA loop per wave (min..max) | barrier | store + add + barrier | MFMAs/wave | loop cycles per MFMA")
for u in range(n_units):
    s = st[u]
    rb, j = (n_units // 4) - 1 - u // 4, u % 4
    loop = s[:, 1] - s[:, 0]
    print(f"{u:3d} {rb:2d} {j} | {s[:, 0].min() - t0:7d} | {loop.min():6d}..{loop.max():6d} | {(s[:, 2] - s[:, 1]).max():6d} | {(s[:, 3] - s[:, 2]).max():6d} | "
          f"{4 * (rb + 1):4d} | {loop.max() / (4 * (rb + 1)):6.1f}")
span = st[:n_units, :, 3].max() - t0
last = st[:n_units, :, 4].max()
wall = st[:n_units, :, 7].max() - w0
print(f"span entry -> last hand-off: {span} cycles; last arriver's epilogue ends at {last - t0} cycles; wall_clock64 span {wall} ticks (100 MHz: {wall * 10} ns)"
      f" -> shader clock ~ {span / (wall * 10e-9) / 1e9:.2f} GHz")
print("workgroup starts (rel cycles), sorted:", np.sort(st[:n_units, :, 0].min(axis=1) - t0)[[0, n_units // 4, n_units // 2, 3 * n_units // 4, n_units - 1]])
if hasattr(amd._lib.lib(), "cf_debug_small_epi"):
    e = (C.c_uint64 * 16)()
    amd._lib.lib().cf_debug_small_epi.argtypes = [C.c_void_p]
    assert amd._lib.lib().cf_debug_small_epi(e) == 0
    e = np.array(e, dtype=np.uint64).astype(np.int64)
    names = ["acquire fence + counter re-arm", "shares -> LDS + barrier", "row-block shares + barrier", "27 adds (+ chi2_extra)", "prior + log P + store",
             "vmcnt(0) (results at the host)", "completion word issued"]
    print("last arriver's epilogue (cycles):", ", ".join(f"{n} {e[k + 1] - e[k]}" for k, n in enumerate(names)))
    print(f"wall_clock64: first workgroup entry -> end of the epilogue {(e[15] - w0) * 10} ns")
