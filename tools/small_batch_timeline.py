#!/usr/bin/env python3
"""Small-batch (W <= 32) synchronous host calls through cf_eval: wall time per call, and -- under
`rocprofv3 --kernel-trace` -- the kernel start / end stamps this script's calls leave (tools/timeline_gaps.py reads them)."""
import importlib, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

pkg = importlib.import_module("cosmology-model-fit_amd")
n_sn = int(os.environ.get("N_SN", "1701"))
syn = pkg.synthetic.pantheon_like(n_sn=n_sn, seed=0)
th = pkg.synthetic.walkers(pkg.sn_pantheon.bounds, 4096, seed=0)
lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
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
