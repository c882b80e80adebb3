"""Phase breakdown of walker_kernel from in-kernel s_memtime stamps (debug build, -DCF_TRSM_STAMPS).
usage (GPU box): make -C cosmology-model-fit_amd/csrc -B EXTRA=-DCF_TRSM_STAMPS && python tools/walker_stamps.py"""
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
buf = (C.c_uint64 * (8 * 8 * 8))()
assert amd._lib.lib().cf_debug_walker_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(8, 8, 8).astype(np.int64)
names = ["theta + cosmology scalars", "E(z), rsqrt, chunk trapezoid (8 nodes / thread)", "wave scan + barrier", "carry + table store + barrier",
         "SN loop (Hermite, log10, store)"]
print("sampled workgroups (one per 512), cycles per phase: min .. max over the 8 waves")
for s in range(8):
    t = st[s]
    if t[0, 0] == 0:
        continue
    row = [f"{(t[:, k + 1] - t[:, k]).min():6d}..{(t[:, k + 1] - t[:, k]).max():6d}" for k in range(5)]
    print(f"wg {s * 512 + 300:4d}: " + " | ".join(row) + f" | total {t[:, 5].max() - t[:, 0].min():6d}")
for k, n in enumerate(names):
    print(f"  phase {k}: {n}")
