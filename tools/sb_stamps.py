"""Phases of small_blocks_kernel from in-kernel s_memtime stamps (debug build, -DCF_TRSM_STAMPS; workgroup 0, thread 0).
usage (GPU box): tools/build_variant.sh stamps -DCF_TRSM_STAMPS && COSMOFIT_LIB=.../libcosmofit_hip_stamps.so WORKLOAD=desi_cmb_des5y:cpl python tools/sb_stamps.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cosmology-model-fit_amd")
fde = "cpl" if os.environ.get("WORKLOAD", "desi_cmb_des5y:cpl").endswith(":cpl") else "lcdm"
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bao_desi_cmb_des5y.npz"))
A = 0.01 * np.random.default_rng(0).standard_normal((g["sigma"].size, 40))
chol = np.linalg.cholesky(np.diag(g["sigma"] ** 2) + A @ A.T)
lk = pkg.likelihoods.DesiCmbDes5y(g["z_cmb"], g["z_hel"], g["obs"], None, g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"], chol=chol, fde=fde)
box = [(-0.5, 0.5), (60.0, 75.0), (0.010, 0.030), (0.01, 0.25), (-4.5, 4.5)] + ([(-3.0, 1.0), (-3.0, 2.0)] if fde == "cpl" else [])
th = pkg.synthetic.walkers(np.array(box), 4096, seed=0)
names = ["theta + cosmology scalars", "pow round 1", "pow round 2, z*, r_d", "Gauss-Legendre loop", "lane sums, CMB vector, 3 x 3 form", "BAO terms",
         "BAO quadratic form"]
for W in [int(w) for w in os.environ.get("WS", "16,4096").split(",")]:
    for _ in range(50):
        lk.log_likelihood(th[:W])
    buf = (C.c_uint64 * 16)()
    lib = pkg._lib.lib()
    lib.cf_debug_sb_stamps.argtypes = [C.c_void_p]
    assert lib.cf_debug_sb_stamps(buf) == 0
    t = np.array(buf, dtype=np.uint64).astype(np.int64)
    print(f"W = {W}, f_DE = {fde}: small_blocks_kernel, workgroup 0, cycles per phase (total {t[7] - t[0]}):")
    for k, n in enumerate(names):
        print(f"  {n:36s} {t[k + 1] - t[k]:7d}")
