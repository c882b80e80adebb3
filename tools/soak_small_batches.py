#!/usr/bin/env python3
"""Soak test of the small-batch paths: thousands of synchronous cf_eval calls of random batch sizes (random offsets into a fixed set of
walkers), every result compared BITWISE with the same walker's result from one 4096-walker batch.  An intermittent hand-off or
completion-word race would show as a mismatch.  WORKLOAD=pantheon (default) | desi_cmb_des5y[:cpl] | desi_cmb (no SN block: the
per-walker kernels + finalize_kernel, whose four waves all store results ahead of the block's completion word);  CALLS (default 6000)."""
import importlib, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

pkg = importlib.import_module("cosmology-model-fit_amd")
wl = os.environ.get("WORKLOAD", "pantheon")
if wl == "pantheon":
    syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
    th = pkg.synthetic.walkers(pkg.sn_pantheon.bounds, 4096, seed=0)
    lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
    f = lk.log_probs_vectorized
elif wl == "desi_cmb":  # bao/desi_cmb.py: BAO + compressed CMB, no SN block -> walker + small blocks + finalize_kernel
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bao_desi_cmb.npz"))
    lk = pkg.likelihoods.DesiCmb(g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"], bounds=g["bounds"])
    th = pkg.synthetic.walkers(np.asarray(g["bounds"]), 4096, seed=0)
    f = lk.log_probs_vectorized
else:
    fde = "cpl" if wl.endswith(":cpl") else "lcdm"
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bao_desi_cmb_des5y.npz"))
    A = 0.01 * np.random.default_rng(0).standard_normal((g["sigma"].size, 40))
    chol = np.linalg.cholesky(np.diag(g["sigma"] ** 2) + A @ A.T)
    lk = pkg.likelihoods.DesiCmbDes5y(g["z_cmb"], g["z_hel"], g["obs"], None, g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"], chol=chol, fde=fde)
    box = [(-0.5, 0.5), (60.0, 75.0), (0.010, 0.030), (0.01, 0.25), (-4.5, 4.5)] + ([(-3.0, 1.0), (-3.0, 2.0)] if fde == "cpl" else [])
    th = pkg.synthetic.walkers(np.array(box), 4096, seed=0)
    f = lk.log_likelihood
ref = np.array(f(th), copy=True)
rng = np.random.default_rng(7)
calls = int(os.environ.get("CALLS", "6000"))
sizes = np.concatenate([rng.integers(1, 200, calls - calls // 10), rng.integers(200, 3000, calls // 10)])
rng.shuffle(sizes)
bad = 0
prev = (0, 0)
t0 = time.perf_counter()
for i, W in enumerate(sizes):
    o = int(rng.integers(0, 4096 - W + 1))
    got = f(th[o:o + W])
    if not np.array_equal(got, ref[o:o + W]):
        bad += 1
        j = int(np.flatnonzero(got != ref[o:o + W])[0])
        elsewhere = np.flatnonzero(ref == got[j])  # a result that belongs to ANOTHER walker says: stale theta / stale table, not arithmetic
        print(f"MISMATCH call {i}: W={W} offset={o} first at {j} ({int((got != ref[o:o + W]).sum())} of {W} wrong): {got[j]!r} vs {ref[o + j]!r}; "
              f"that value is walker {elsewhere.tolist()}'s; previous call: W={prev[0]} offset={prev[1]}", flush=True)
        if bad > 5:
            break
    prev = (int(W), o)
print(f"{wl}: {len(sizes)} calls, sizes 1..{int(sizes.max())}, {bad} mismatches, {time.perf_counter() - t0:.1f} s")
lk.engine.close()
sys.exit(1 if bad else 0)
