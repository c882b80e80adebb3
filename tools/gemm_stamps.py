"""Per-workgroup timeline of the inverse-GEMM solve from in-kernel s_memtime stamps (debug build, -DCF_TRSM_STAMPS).
usage (GPU box): make -C cosmology-model-fit_amd/csrc -B EXTRA=-DCF_TRSM_STAMPS && python tools/gemm_stamps.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("cosmology-model-fit_amd")
syn = amd.synthetic
data = syn.pantheon_like(int(os.environ.get("N_SN", 1701)), seed=0)
lk = amd.sn_pantheon.PantheonLikelihood(data["z_cmb"], data["z_hel"], data["obs"], chol=data["chol"], latency_mode=True)
th = syn.walkers(lk.bounds, 4096, seed=1)
for _ in range(3):
    lk.log_probability(th)
buf = (C.c_uint64 * (64 * 4 * 4))()
assert amd._lib.lib().cf_debug_gemm_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(64, 4, 4).astype(np.int64)
nrb = int((st[:, 0, 0] > 0).sum())
t0 = st[:nrb, :, 0].min()
print("rb | start (rel) | prologue (max over waves) | loop | MFMAs/wave | cyc per MFMA in loop | epilogue | total")
for rb in range(nrb - 1, -1, -1):
    s = st[rb]
    n_mfma = 2 * (rb + 1) * 16
    loop = s[:, 2] - s[:, 1]
    print(f"{rb:2d} | {s[:, 0].min() - t0:8d} | {(s[:, 1] - s[:, 0]).max():6d} | {loop.min():7d}..{loop.max():7d} | {n_mfma:5d} | {loop.mean() / n_mfma:6.1f} | "
          f"{(s[:, 3] - s[:, 2]).max():6d} | {s[:, 3].max() - s[:, 0].min():7d}")
print("span of this panel's workgroups:", st[:nrb, :, 3].max() - t0)
