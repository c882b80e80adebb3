import importlib, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("cosmology-model-fit_amd")
syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
th = pkg.synthetic.walkers(pkg.sn_pantheon.bounds, 4096, seed=0)
for G in (4000, 3300, 2900, 2400, 2000, 1000):
    pkg.sn_pantheon.N_GRID = G
    lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
    for _ in range(3): lk.log_probs_vectorized(th)
    lk.engine.enable_timing(32)
    for _ in range(20): lk.log_probs_vectorized(th)
    k = np.array(lk.engine.kernel_ms()[-15:])
    print(f"G={G}: LDS {(G + G // 8 + 2) * 16 / 1024:.0f} KB  walker {k[:,0].mean()*1e3:.1f} us  solve {k[:,1].mean()*1e3:.1f} us")
    lk.engine.close()
