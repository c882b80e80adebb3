#!/usr/bin/env python3
"""How much does the chip gain when walker_kernel (FP64 VALU) and the solve (FP64 matrix cores) of INDEPENDENT
evaluations run side by side?  Two engines on two streams, K evaluations each, against the same 2K evaluations on
one stream.  Upper bound for any overlap scheme inside one evaluation."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cosmology-model-fit_amd")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 100
syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
lks = [pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"]) for _ in range(2)]
th = torch.from_numpy(pkg.synthetic.walkers(pkg.sn_pantheon.bounds, W, seed=0)).cuda()
outs = [torch.empty(W, dtype=torch.float64, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(two_streams, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(2):
            s = streams[i if two_streams else 0]
            lks[i].engine.eval_device(th.data_ptr(), W, outs[i].data_ptr(), pkg.CF_OUT_LOGP, s.cuda_stream)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for mode in (False, True, False, True):
    run(mode, 5)
    dt = run(mode, K)
    print(f"W={W} two_streams={mode}: {2 * K * W / dt:.4e} evals/s, {dt / (2 * K) * 1e3:.4f} ms per evaluation")
