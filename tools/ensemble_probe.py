#!/usr/bin/env python3
"""Time per ensemble step of the device-resident sampler, per move, next to the bare likelihood evaluation."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("cosmology-model-fit_amd")
W = int(os.environ.get("W", 4096))
syn = amd.synthetic.pantheon_like(n_sn=1701, seed=0)
lk = amd.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
start = amd.synthetic.THETA_TRUE + np.array([0.02, 1.0, 0.03, 0.3]) * np.random.default_rng(1).standard_normal((W, 4))
x0 = torch.from_numpy(start).to("cuda:0")
f = lk.engine.torch_log_prob()
for _ in range(3): f(x0[: W // 2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): f(x0[: W // 2])
torch.cuda.synchronize(); print(f"log P of one half ({W // 2} walkers): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
for name in ("stretch", "de", "kde"):
    e = amd.ensemble.ShardedEnsemble(f, x0, seed=3, moves=((name, 1.0),))
    e.run(5); torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run(30); torch.cuda.synchronize()
    n_upd = 3 if name == "de" else 2
    print(f"{name:8s}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step ({n_upd} split updates), acceptance {e.acceptance_fraction():.2f}")
e = amd.ensemble.ShardedEnsemble(f, x0, seed=3, moves=amd.ensemble.REFERENCE_MOVES)
e.run(5); torch.cuda.synchronize(); t0 = time.perf_counter()
e.run(100); torch.cuda.synchronize()
print(f"reference mixture (KDE 30 % + DE 70 %): {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per step")
t0 = time.perf_counter()
for _ in range(100): e._pick_move()
print(f"host side of one step: _pick_move {(time.perf_counter() - t0) / 100 * 1e6:.0f} us")
