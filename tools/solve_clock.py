#!/usr/bin/env python3
"""In-kernel timeline of the throughput solve kernel from the DIAGNOSTIC build (tools/build_variant.sh clock -DCF_DIAG_CLOCK):
    COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so python tools/solve_clock.py [W ...]
Per batch size: the kernel's span on the 100 MHz real-time counter, when the workgroups start and hand off, how much of a wave's
life is spent inside K loops, how long the last arrivers' epilogues take, when each CU runs out of work, and the shader clock the chip holds (d s_memtime / d s_memrealtime x 100 MHz, median over workgroups)
after >= 2 s of back-to-back evaluations."""
import ctypes as C, importlib, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

pkg = importlib.import_module("cosmology-model-fit_amd")
lib = pkg._lib.lib()
lib.cf_debug_solve_clock.argtypes = [C.c_void_p]
lib.cf_debug_solve_kloop.argtypes = [C.c_void_p]
lib.cf_debug_solve_phase.argtypes = [C.c_void_p]
NP = 2
syn = pkg.synthetic.pantheon_like(n_sn=1701, seed=0)
lk = pkg.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"])
import torch

results = {}
for W in [int(a) for a in sys.argv[1:]] or [4096]:
    assert W <= 4096, "the stamp buffers hold 4096 workgroups"
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
    last = b[:, 7] > 0  # workgroups that arrived last for their panel and ran its epilogue
    start, handoff = (b[:, 1] - rt0) / 100.0, (b[:, 3] - rt0) / 100.0  # us
    end = np.where(last, (b[:, 5] - rt0) / 100.0, handoff)
    clk = (b[:, 2] - b[:, 0]) / np.maximum(b[:, 3] - b[:, 1], 1) * 0.1  # GHz
    q = lambda x: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(x, [0, 10, 50, 90, 100]))
    print(f"W = {W}: {n} workgroups (one unit each), kernel span {end.max():.1f} us (real-time counter)")
    print("  workgroup start  [us]:", q(start))
    print("  workgroup exit   [us]:", q(end))
    print("  last arrivers' epilogue [us]:", q((end - handoff)[last]), f"({int(last.sum())} panels)")
    kb = np.zeros(4096 * 4, dtype=np.uint64)
    assert lib.cf_debug_solve_kloop(kb.ctypes.data) == 0
    kb = kb.reshape(4096, 4).astype(np.int64)[:n]
    life = b[:, 2] - b[:, 0]  # shader cycles from entry to the end of the hand-off
    mfmas = kb[:, 1] * 8 * (2 if W > 512 else 1)  # per wave: 2 K-steps x 4 tiles x NP per pair
    print("  wave 0: share of the workgroups' lifetime inside K loops: %.3f;  cycles per MFMA inside K loops: %.1f;  outside K loops per unit: %.0f cycles"
          % (kb[:, 0].sum() / life.sum(), kb[:, 0].sum() / mfmas.sum(), (life.sum() - kb[:, 0].sum()) / n))
    ph = np.zeros(4096 * 16, dtype=np.uint64)
    assert lib.cf_debug_solve_phase(ph.ctypes.data) == 0
    ph = ph.reshape(4096, 16).astype(np.int64)[:n]
    kb_, ke_ = ph[:, 0:8:2], ph[:, 1:8:2]  # K loop begin / end of the four waves
    t0 = b[:, 0]
    m = lambda x: "%.0f" % np.mean(x)
    print("  per unit, mean shader cycles: entry -> K loop begins (wave 0) %s | K loop (mean over waves) %s | first -> last wave out of the K loop %s | "
          "last wave out -> first exchange barrier passed %s | -> shares formed %s | -> stores acknowledged %s | -> arrival add returned, barrier %s"
          % (m(kb_[:, 0] - t0), m((ke_ - kb_).mean(axis=1)), m(ke_.max(axis=1) - ke_.min(axis=1)), m(ph[:, 8] - ke_.max(axis=1)), m(ph[:, 9] - ph[:, 8]),
             m(ph[:, 10] - ph[:, 9]), m(ph[:, 11] - ph[:, 10])))
    print("  spread of the four waves' K-loop BEGIN: %s;  K-loop duration, slowest - fastest wave: %s" % (m(kb_.max(axis=1) - kb_.min(axis=1)), m((ke_ - kb_).max(axis=1) - (ke_ - kb_).min(axis=1))))
    long_ = life > np.percentile(life, 50)
    print("  shader clock [GHz] (workgroups of the longer half):", "min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f" % tuple(np.percentile(clk[long_], [0, 10, 50, 90, 100])))
    # per CU (XCC_ID, and SE / SH / CU of HW_ID): when its last workgroup is done; what the chip loses to the ragged end
    hw = b[:, 6]
    cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff)  # xcc, [se_id 15:13, sh_id 12, cu_id 11:8]
    ids, inv = np.unique(cu, return_inverse=True)
    cu_end = np.zeros(len(ids)); np.maximum.at(cu_end, inv, handoff)
    busy_wg = np.zeros(len(ids)); np.add.at(busy_wg, inv, handoff - start)
    print(f"  {len(ids)} CUs, workgroups per CU: min {np.bincount(inv).min()} max {np.bincount(inv).max()}")
    print("  CU's LAST unit is handed off [us]:", q(cu_end))
    print("  idle CU time before the kernel's last unit ends: %.1f us mean per CU = %.1f %% of the span;  mean workgroups resident per CU over the span: %.2f"
          % ((cu_end.max() - cu_end).mean(), 100 * (cu_end.max() - cu_end).mean() / end.max(), busy_wg.sum() / len(ids) / end.max()))
    # the first 1024 workgroups: does block b sit with b + 256, b + 512, b + 768 (static placement of a resident grid)?
    same = np.mean([cu[i] == cu[i + 256] for i in range(min(256, n - 256))]) if n > 256 else float("nan")
    print("  blocks b and b + 256 on the same CU: %.0f %%" % (100 * same))
    results[W] = {"kernel_span_us": float(end.max()), "clock_ghz_median": float(np.median(clk[long_])), "clock_ghz_p10_p90": [float(x) for x in np.percentile(clk[long_], [10, 90])],
                  "share_of_wave_life_in_k_loops": float(kb[:, 0].sum() / life.sum()), "cycles_per_mfma_in_k_loops": float(kb[:, 0].sum() / mfmas.sum()),
                  "idle_cu_fraction_at_the_end": float((cu_end.max() - cu_end).mean() / end.max()), "mean_workgroups_resident_per_cu": float(busy_wg.sum() / len(ids) / end.max())}
lk.engine.close()
if os.environ.get("OUT_JSON"):
    import json, re
    bare = None
    if os.environ.get("BARE_TXT") and os.path.exists(os.environ["BARE_TXT"]):  # tools/coexec_f64_rate output of the same box
        bare = {m.group(1) + " waves/SIMD": {"tflops": float(m.group(2)), "clock_ghz": float(c)} for c, m in
                ((re.search(r"clock ([\d.]+) GHz", l).group(1), re.search(r"MFMA ([\d.]+) waves/SIMD: ([\d.]+) TFLOP/s", l))
                 for l in open(os.environ["BARE_TXT"]) if "roles=0" in l and "MFMA" in l)}
    w = 4096 if 4096 in results else max(results)
    json.dump({"method": "diagnostic build (-DCF_DIAG_CLOCK): s_memtime / s_memrealtime x 100 MHz per workgroup of tri_gemm_chi2_kernel after >= 2 s of "
                         "back-to-back evaluations, median over the longer half of the workgroups; bare loop: tools/coexec_f64_rate on the same box",
               "solve_kernel_clock_ghz_median": results[w]["clock_ghz_median"], "walkers": w, "by_batch_size": {str(k): v for k, v in results.items()},
               "bare_mfma_loop_tflops": bare}, open(os.environ["OUT_JSON"], "w"), indent=1)
