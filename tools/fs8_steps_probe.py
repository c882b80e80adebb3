#!/usr/bin/env python3
"""growth_kernel time against the number of RK4 steps (256 lanes per walker, 1 / 2 / 4 / 8 steps per lane): splits the per-step
cost from the fixed cost (scan, data points, quadratic form).  fs8/fs8.py shape, 4096 walkers, device-resident theta."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cosmology-model-fit_amd")
import torch
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fs8_fs8.npz"))
th = torch.from_numpy(pkg.synthetic.walkers(g["bounds"], 4096, seed=1)).cuda()
out = torch.empty(4096, dtype=torch.float64, device="cuda")
for steps in (256, 512, 1024, 2048):
    lk = pkg.likelihoods.Fs8(g["fs8_z"], g["fs8_val"], g["fs8_cov"], None, fid=g["fs8_fid"], bounds=g["bounds"], steps=steps)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        lk.engine.eval_device(th.data_ptr(), 4096, out.data_ptr(), pkg.CF_OUT_LOGP, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        lk.engine.eval_device(th.data_ptr(), 4096, out.data_ptr(), pkg.CF_OUT_LOGP, st)
    torch.cuda.synchronize()
    print(f"steps {steps:5d}: {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us per evaluation (walker + growth + finalize kernels)")
    lk.engine.close()
