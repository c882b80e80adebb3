#!/usr/bin/env python3
"""Does walker_kernel gain from a third workgroup per CU?  Its LDS table (16 B per grid node + skew, + 8 KB static) lets 2 workgroups
share a CU at the reference's G = 4000; below ~2570 nodes 3 fit.  Time per 4096 walkers on either side of that boundary."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cosmology-model-fit_amd")
syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
th = pkg.synthetic.walkers(pkg.sn_pantheon.bounds, 4096, seed=0)
P = pkg.Param
for G in (4000, 3400, 2800, 2700, 2600, 2550, 2500, 2400, 2000):
    eng = pkg.LikelihoodEngine(ndim=4, z_max=syn["z_max"], n_grid=G, params=dict(offset=P(0), H0=P(1), Om=P(2), v=P(3)),
                               sn=dict(z_cmb=syn["z_cmb"], z_hel=syn["z_hel"], obs=syn["obs"], chol=syn["chol"]),
                               bounds=pkg.sn_pantheon.bounds)
    for _ in range(30):
        eng.log_probability(th)
    eng.enable_timing(64)
    for _ in range(64):
        eng.log_probability(th)
    k = np.array(eng.kernel_ms())
    lds = (G + (G >> 3) + 2) * 16 + 8192
    print(f"G = {G:5d}: LDS per workgroup {lds / 1024:5.1f} KB -> {min(int(160 * 1024 // lds), 3)} per CU (3 by registers); walker_kernel {k[:, 0].mean() * 1e3:6.1f} us")
    eng.close()
