#!/usr/bin/env python3
"""In-kernel timeline of the throughput solve kernel from the DIAGNOSTIC build (tools/build_variant.sh clock -DCF_DIAG_CLOCK):
    COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so python tools/solve_clock.py [W ...]
Per batch size: the kernel's span on the 100 MHz real-time counter, when the workgroups leave their unit loops, how long the
deferred panel epilogues take, and the shader clock the chip holds (d s_memtime / d s_memrealtime x 100 MHz, median over workgroups)
after >= 2 s of back-to-back evaluations."""
import ctypes as C, importlib, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

pkg = importlib.import_module("cosmology-model-fit_amd")
lib = pkg._lib.lib()
lib.cf_debug_solve_clock.argtypes = [C.c_void_p]
lib.cf_debug_solve_kloop.argtypes = [C.c_void_p]
NP = 2
syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
import torch

for W in [int(a) for a in sys.argv[1:]] or [4096]:
    th = torch.from_numpy(pkg.synthetic.walkers(pkg.sn_pantheon.bounds, W, seed=0)).cuda()
    out = torch.empty(W, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:  # the sustained clock, not the ramp
        for _ in range(32):
            lk.engine.eval_device(th.data_ptr(), W, out.data_ptr(), pkg.CF_OUT_LOGP, st)
        torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    assert lib.cf_debug_solve_clock(buf.ctypes.data) == 0
    b = buf.reshape(4096, 8).astype(np.int64)
    b = b[b[:, 1] > 0]
    n = len(b)
    rt0 = b[:, 1].min()
    start, loop_end, end = (b[:, 1] - rt0) / 100.0, (b[:, 3] - rt0) / 100.0, (b[:, 5] - rt0) / 100.0  # us
    clk = (b[:, 4] - b[:, 0]) / np.maximum(b[:, 5] - b[:, 1], 1) * 0.1  # GHz
    q = lambda x: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(x, [0, 10, 50, 90, 100]))
    print(f"W = {W}: {n} workgroups, kernel span {end.max():.1f} us (real-time counter)")
    print("  workgroup start  [us]:", q(start))
    print("  unit loop left   [us]:", q(loop_end))
    print("  workgroup exit   [us]:", q(end))
    print("  epilogue time of workgroups with panels [us]:", q((end - loop_end)[b[:, 7] > 0]) if (b[:, 7] > 0).any() else "-")
    print("  panels per workgroup:", dict(zip(*np.unique(b[:, 7], return_counts=True))))
    kb = np.zeros(4096 * 4, dtype=np.uint64)
    assert lib.cf_debug_solve_kloop(kb.ctypes.data) == 0
    kb = kb.reshape(4096, 4).astype(np.int64)[:n]
    life = b[:, 2] - b[:, 0]  # shader cycles from entry to the end of the unit loop
    mfmas = kb[:, 1] * 8 * (2 if W > 512 else 1)  # per wave: 2 K-steps x 4 tiles x NP per pair
    print("  wave 0: share of the unit-loop lifetime inside K loops: %.3f;  cycles per MFMA inside K loops: %.1f (x %d workgroups per CU: pipe busy %.3f there);  outside K loops per unit: %.0f cycles;  units per workgroup %.1f"
          % (kb[:, 0].sum() / life.sum(), kb[:, 0].sum() / mfmas.sum(), n // 256, 64.0 * (n // 256) / (kb[:, 0].sum() / mfmas.sum()),
             (life.sum() - kb[:, 0].sum()) / kb[:, 2].sum(), kb[:, 2].mean()))
    print("  shader clock [GHz]:", "min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f" % tuple(np.percentile(clk, [0, 10, 50, 90, 100])))
    # per CU (XCC_ID, and SE / SH / CU of HW_ID): when its last workgroup left the unit loop; what the chip loses to the ragged end
    hw = b[:, 6]
    cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff)  # xcc, [se_id 15:13, sh_id 12, cu_id 11:8]
    ids, inv = np.unique(cu, return_inverse=True)
    cu_end = np.zeros(len(ids)); np.maximum.at(cu_end, inv, loop_end)
    cu_first = np.full(len(ids), 1e30); np.minimum.at(cu_first, inv, loop_end)
    per_cu = np.bincount(inv)
    print(f"  {len(ids)} CUs, workgroups per CU {dict(zip(*np.unique(per_cu, return_counts=True)))}")
    print("  CU's LAST workgroup leaves the unit loop [us]:", q(cu_end))
    print("  CU's FIRST workgroup leaves the unit loop [us]:", q(cu_first))
    print("  idle CU time before the kernel's last unit ends: %.1f us mean per CU = %.1f %% of the span" % ((cu_end.max() - cu_end).mean(), 100 * (cu_end.max() - cu_end).mean() / end.max()))
lk.engine.close()
