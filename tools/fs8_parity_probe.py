#!/usr/bin/env python3
"""Growth block: what the GPU achieves against the fixtures' converged theory (theory_tight) and against chi^2 recomputed on the
host from it -- the numbers tests/test_fs8.py's bars are set from.  a_grid = N (the scripts' PCHIP on their log grid) vs 0 (direct)."""
import importlib, os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
amd = importlib.import_module("cosmology-model-fit_amd")
from conftest import golden
import test_fs8 as T
from oracle import oracle_np as onp

L = amd.likelihoods
build = {
    "fs8_fs8": lambda g, **kw: L.Fs8(g["fs8_z"], g["fs8_val"], g["fs8_cov"], None, fid=g["fs8_fid"], bounds=g["bounds"], **kw),
    "bao_desi_cmb_union3_fs8": lambda g, **kw: L.DesiCmbUnion3Fs8(g["z_cmb"], g["z_hel"], g["obs"], g["cov_sn"], g["bao_z"], g["bao_val"], g["bao_qty"],
                                                                   g["bao_inv_cov"], g["fs8_z"], g["fs8_val"], g["fs8_cov"], g["fs8_fid"], **kw),
    "ohd_cc_fs8": lambda g, **kw: L.CcFs8(g["cc_z"], g["cc_h"], g["cc_cov"], g["fs8_z"], g["fs8_val"], g["fs8_cov"], g["fs8_fid"], **kw),
    "fs8_fs8_cmb": lambda g, **kw: L.Fs8Cmb(g["fs8_z"], g["fs8_val"], g["fs8_cov"], g["fs8_fid"], bounds=g["bounds"], **kw),
    "bao_desi_fs_lya_cc_fs8": lambda g, **kw: L.DesiFsLyaCcFs8(g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"], g["cc_z"], g["cc_h"], g["cc_cov"],
                                                                 g["fs8_z"], g["fs8_val"], g["fs8_cov"], g["fs8_fid"], **kw),
}
for name, mk in build.items():
    g = golden(name)
    olk = T.CASES[name](g)
    for steps in (0, 2048):
        lk = mk(g, steps=steps)
        nt = len(g["theory_tight"])
        th = g["thetas"][:nt]
        got = np.array([lk.fs8_theory(t) for t in th])
        e_tight = np.max(np.abs(got / g["theory_tight"] - 1))
        e_ref = np.max(np.abs(got / g["theory"] - 1))
        parts = lk.engine.parts(th)
        c_tight = np.array([T.chi2_fs8_from_theory(olk, t, g["theory_tight"][k]) for k, t in enumerate(th)])
        e_chi = np.max(np.abs(parts["chi2_fs8"] / c_tight - 1))
        print(f"{name:26s} steps {steps or 1024:4d}: theory vs tight {e_tight:.2e}  vs reference {e_ref:.2e}  chi2_fs8 vs host(theory_tight) {e_chi:.2e}", flush=True)
        lk.engine.close()
