#!/usr/bin/env python3
"""Synchronous host calls (cf_eval, numpy in / out) vs batch size with and without in-place access to the pinned staging block
(CF_ZEROCOPY_MAX); no timing events (they cost launches of their own).  Run once per setting: the knob is read once."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cosmology-model-fit_amd")
import numpy as np
syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
th = pkg.synthetic.walkers(pkg.sn_pantheon.bounds, 32768, seed=0)
lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
ref = lk.log_probs_vectorized(th)
print("CF_ZEROCOPY_MAX =", os.environ.get("CF_ZEROCOPY_MAX", "(default)"))
for W in [int(w) for w in os.environ.get("WS", "1,32,75,150,256,512,1024,2048,4096,8192,16384,32768").split(",")]:
    for _ in range(5): got = lk.log_probs_vectorized(th[:W])
    assert np.array_equal(got, ref[:W])
    t0 = time.perf_counter()
    reps = 200 if W <= 4096 else 60
    for _ in range(reps): lk.log_probs_vectorized(th[:W])
    dt = (time.perf_counter() - t0) / reps * 1e6
    print(f"W={W:5d}: wall {dt:7.1f} us per call ({W / dt:7.3f} evals/us)")
