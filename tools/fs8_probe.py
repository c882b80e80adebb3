#!/usr/bin/env python3
"""Time of one evaluation of the growth-rate likelihoods (fs8/fs8.py and bao/desi_cmb_union3_fs8.py shapes), 4096 walkers."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cosmology-model-fit_amd")
G = lambda n: np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", n + ".npz"))
g = G("fs8_fs8")
lk = pkg.likelihoods.Fs8(g["fs8_z"], g["fs8_val"], g["fs8_cov"], None, fid=g["fs8_fid"], bounds=g["bounds"])
th = pkg.synthetic.walkers(g["bounds"], 4096, seed=1)
lk.log_probs_vectorized(th)
t0 = time.perf_counter()
for _ in range(20):
    lk.log_probs_vectorized(th)
print(f"fs8/fs8.py shape (56 data, growth ODE only), 4096 walkers, host buffers: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
g = G("bao_desi_cmb_union3_fs8")
lk2 = pkg.likelihoods.DesiCmbUnion3Fs8(g["z_cmb"], g["z_hel"], g["obs"], g["cov_sn"], g["bao_z"], g["bao_val"], g["bao_qty"],
                                       g["bao_inv_cov"], g["fs8_z"], g["fs8_val"], g["fs8_cov"], g["fs8_fid"])
box = np.array([(-1.0, 1.0), (50.0, 90.0), (0.01, 0.03), (0.05, 0.25), (-8.0, 8.0), (0.5, 1.1)])
th = pkg.synthetic.walkers(box, 4096, seed=1)
lk2.log_likelihood(th)
t0 = time.perf_counter()
for _ in range(20):
    lk2.log_likelihood(th)
print(f"bao/desi_cmb_union3_fs8.py shape (22 SN bins + 13 BAO + CMB + 56 growth data, physical E(z)), 4096 walkers: "
      f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
