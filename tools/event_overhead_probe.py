#!/usr/bin/env python3
"""What do the per-evaluation HIP events cost?  K back-to-back evaluations on one stream with kernel timing off / on / sampled."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cosmology-model-fit_amd")
W, K = 4096, 200
syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
th = torch.from_numpy(pkg.synthetic.walkers(pkg.sn_pantheon.bounds, W, seed=0)).cuda()
out = torch.empty(W, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        lk.engine.eval_device(th.data_ptr(), W, out.data_ptr(), pkg.CF_OUT_LOGP, s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for label, slots, stride in (("timing off", 0, 1), ("timing on, every call", K, 1), ("timing off", 0, 1), ("timing on, every 8th call", K, 8),
                             ("timing on, every call", K, 1), ("timing on, every 8th call", K, 8), ("timing off", 0, 1)):
    lk.engine.enable_timing(slots, stride) if slots else lk.engine.enable_timing(0)
    run(20)
    print(f"{label:28s}: {run(K):.4f} ms per evaluation")
