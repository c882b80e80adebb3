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
if os.environ.get("WORKLOAD", "pantheon") == "pantheon":
    data = syn.pantheon_like(int(os.environ.get("N_SN", 1701)), seed=0)
    lk = amd.sn_pantheon.PantheonLikelihood(data["z_cmb"], data["z_hel"], data["obs"], chol=data["chol"])
    th = syn.walkers(lk.bounds, 4096, seed=1)
    for _ in range(3):
        lk.log_probability(th)
else:  # the joint likelihood of bao/desi_cmb_des5y.py on the committed fixture data (as bench.py --workload desi_cmb_des5y)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bao_desi_cmb_des5y.npz"))
    A = 0.01 * np.random.default_rng(0).standard_normal((g["sigma"].size, 40))
    chol = np.linalg.cholesky(np.diag(g["sigma"] ** 2) + A @ A.T)
    fde = os.environ.get("FDE", "lcdm")
    lk = amd.likelihoods.DesiCmbDes5y(g["z_cmb"], g["z_hel"], g["obs"], None, g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"], chol=chol,
                                      fde=fde)
    box = [(-0.5, 0.5), (60.0, 75.0), (0.010, 0.030), (0.01, 0.25), (-4.5, 4.5)] + ([(-3.0, 1.0), (-3.0, 2.0)] if fde == "cpl" else [])
    box = np.array(box)
    th = syn.walkers(box, 4096, seed=1)
    for _ in range(3):
        lk.log_likelihood(th)
buf = (C.c_uint64 * (8 * 8 * 8))()
assert amd._lib.lib().cf_debug_walker_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(8, 8, 8).astype(np.int64)
names = ["theta + cosmology scalars", "E(z), rsqrt, chunk trapezoid (8 nodes / thread)", "wave scan + barrier", "carry + table store + barrier",
         "SN loop (Hermite, log10, store)", "z* / r_drag powers, CC + barrier", "BAO, Gauss-Legendre nodes, quadratic forms, combiner"]
print("sampled workgroups (one per 512), cycles per phase: min .. max over the 8 waves")
for s in range(8):
    t = st[s]
    if t[0, 0] == 0:
        continue
    last = 7 if t[0, 7] > 0 else 5
    row = [f"{(t[:, k + 1] - t[:, k]).min():6d}..{(t[:, k + 1] - t[:, k]).max():6d}" for k in range(last)]
    print(f"wg {s * 512 + 300:4d}: " + " | ".join(row) + f" | total {t[:, last].max() - t[:, 0].min():6d}")
for k, n in enumerate(names):
    print(f"  phase {k}: {n}")
