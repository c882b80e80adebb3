#!/usr/bin/env python3
"""Small-batch (W <= 32) synchronous host calls through cf_eval: wall time per call, and -- under
`rocprofv3 --kernel-trace` -- the kernel start / end stamps this script's calls leave (tools/timeline_gaps.py reads them)."""
import importlib, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

pkg = importlib.import_module("cosmology-model-fit_amd")
n_sn = int(os.environ.get("N_SN", "1701"))
if os.environ.get("WORKLOAD", "pantheon") == "pantheon":
    syn = pkg.synthetic.pantheon_like(n_sn=n_sn, seed=0)
    th = pkg.synthetic.walkers(pkg.sn_pantheon.bounds, 4096, seed=0)
    lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
else:  # WORKLOAD=desi_cmb_des5y[:cpl]: the joint likelihood of bench.py --workload desi_cmb_des5y [--fde cpl] (BASELINE configs[2] shape)
    fde = "cpl" if os.environ["WORKLOAD"].endswith(":cpl") else "lcdm"
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bao_desi_cmb_des5y.npz"))
    A = 0.01 * np.random.default_rng(0).standard_normal((g["sigma"].size, 40))
    chol = np.linalg.cholesky(np.diag(g["sigma"] ** 2) + A @ A.T)
    lk = pkg.likelihoods.DesiCmbDes5y(g["z_cmb"], g["z_hel"], g["obs"], None, g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"],
                                      chol=chol, fde=fde)
    box = [(-0.5, 0.5), (60.0, 75.0), (0.010, 0.030), (0.01, 0.25), (-4.5, 4.5)] + ([(-3.0, 1.0), (-3.0, 2.0)] if fde == "cpl" else [])
    th = pkg.synthetic.walkers(np.array(box), 4096, seed=0)
    lk.log_probs_vectorized = lk.log_likelihood
lk.log_probs_vectorized(th)
ref = lk.log_probs_vectorized(th[:64])
for W in [int(w) for w in os.environ.get("WS", "1,16,32,64").split(",")]:
    for _ in range(50):
        out = lk.log_probs_vectorized(th[:W])
    t0 = time.perf_counter()
    reps = int(os.environ.get("REPS", "400"))
    for _ in range(reps):
        out = lk.log_probs_vectorized(th[:W])
    dt = (time.perf_counter() - t0) / reps * 1e6
    assert np.array_equal(out, ref[:W]) or W > 64 or os.environ.get("SKIP_CHECK"), "a walker's result must not depend on the batch size"
    print(f"W={W:4d}: {dt:6.1f} us per synchronous cf_eval call ({W / dt:6.3f} evals/us)", flush=True)
lk.engine.close()
