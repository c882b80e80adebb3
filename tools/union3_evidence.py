#!/usr/bin/env python3
"""Laplace log-evidence of sn/union3_1.py (real data) on the GPU engine, next to the nautilus value the reference publishes
(sn/union3_1.py:155-168: log Z = -20.5 with the velocity step, -21.9 without)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("cosmology-model-fit_amd")
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "sn_union3_1.npz"))
box = amd.likelihoods.SnUnion3.PRIOR_BOX
lk = amd.likelihoods.SnUnion3(g["z_cmb"], g["z_hel"], g["obs"], g["cov"], H0=float(g["H0"]), bounds=box)
rng = np.random.default_rng(3)
start = np.array([0.0, 0.3, -3.0]) + np.array([0.02, 0.02, 1.0]) * rng.standard_normal((2048, 3))
ens = amd.ensemble.ShardedEnsemble(lk.engine.torch_log_prob(), torch.from_numpy(start).to("cuda:0"), seed=9, moves=amd.ensemble.REFERENCE_MOVES)
ens.run(400)
samples, logp = ens.x.cpu().numpy(), ens.logp.cpu().numpy()
out = amd.laplace.log_evidence(samples, logp, lk.log_probs_vectorized, box)
print("Laplace log evidence:", out if np.ndim(out) == 0 else out)
