"""Phase breakdown of the blocked solve from in-kernel s_memtime stamps (debug build, -DCF_TRSM_STAMPS).
usage (GPU box): make -C cosmology-model-fit_amd/csrc -B EXTRA=-DCF_TRSM_STAMPS && python tools/trsm_stamps.py"""
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
for _ in range(3):
    lk.log_probability(th)
buf = (C.c_uint64 * (16 * 16 * 5))()
rc = amd._lib.lib().cf_debug_trsm_stamps(buf)
assert rc == 0, rc
st = np.array(buf, dtype=np.uint64).reshape(16, 16, 5).astype(np.int64)
nw = int((st[:, 0, 0] > 0).sum())
nb = int((st[0, :, 0] > 0).sum())
t0 = st[:nw, 0, 0].min()
print(f"waves {nw} block rows {nb}; cycles relative to the first stamp (s_memtime ticks)")
print("b | update (max over waves) | publish+barrier | diag (min..max) | barrier wait | total")
for b in range(nb):
    s = st[:nw, b, :]
    upd = s[:, 1] - s[:, 0]
    pub = s[:, 2] - s[:, 1]
    dia = s[:, 3] - s[:, 2]
    bar = s[:, 4] - s[:, 3]
    tot = s[:, 4].max() - s[:, 0].min()
    print(f"{b} | {upd.min():7d}..{upd.max():7d} | {pub.min():6d}..{pub.max():6d} | {dia.min():6d}..{dia.max():6d} | {bar.min():6d}..{bar.max():6d} | {tot:7d}")
print("whole loop:", st[:nw, nb - 1, 4].max() - t0)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "trsm_stamps.npy"), st)
