#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter CSVs per kernel (one row per dispatch and counter)."""
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, cs in acc.items():
    if not ("trsm" in k or "tri_gemm" in k or "walker" in k or "small_blocks" in k):
        continue
    lines.append(f"== {k[:60]}  (dispatches: {max(len(v) for v in cs.values())})")
    for c, v in sorted(cs.items()):
        v = v[2:] if len(v) > 3 else v  # drop warm-up dispatches
        lines.append(f"   {c:36s} {sum(v)/len(v):18.1f}")
# durations of the SAME dispatches the counters were read on (every pass also carries --kernel-trace): the counters and the clock
# come from one run, so busy cycles, duration and clock can be set against each other
dur = collections.defaultdict(dict)  # kernel -> pass -> mean duration in us
for d in sorted(glob.glob(os.path.join(out, "pass*"))):
    if not os.path.isdir(d):
        continue
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for k, v in per.items():
        v = v[2:] if len(v) > 3 else v
        dur[k][os.path.basename(d)] = sum(v) / len(v)
roof = {}
for k, cs in acc.items():
    if "tri_gemm" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    mean = lambda c: (lambda v: sum(v) / len(v))(cs[c][2:] if len(cs[c]) > 3 else cs[c])
    d_all = dur.get(k, {})
    d_us = sum(d_all.values()) / len(d_all) if d_all else None
    busy, gui = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE")
    roof = {"kernel": "tri_gemm_chi2_kernel", "duration_us_under_pmc_by_pass": d_all, "duration_us_under_pmc": d_us,
            "SQ_VALU_MFMA_BUSY_CYCLES": busy, "mfma_instructions": busy / 64.0, "GRBM_GUI_ACTIVE": gui,
            "clock_ghz_from_gui_active": gui / 8.0 / (d_us * 1e3) if d_us else None,  # reads high on dispatches < 0.3 ms (MI355X_MICROARCH.md)
            "mfma_busy_vs_gui_active": busy / 1024.0 / (gui / 8.0)}
    lines.append(f"== roofline evidence, {k[:40]}: duration under PMC {d_us and round(d_us, 1)} us (per pass {({p: round(v, 1) for p, v in d_all.items()})}); "
                 f"MFMAs {busy / 64:.0f}; busy / (GUI_ACTIVE / 8) = {roof['mfma_busy_vs_gui_active']:.3f}; GUI_ACTIVE / 8 / duration = "
                 f"{roof['clock_ghz_from_gui_active'] and round(roof['clock_ghz_from_gui_active'], 3)} GHz")
if roof:
    import json as _json
    _json.dump(roof, open(os.path.join(out, "pmc_roofline.json"), "w"), indent=1)
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "pmc_summary.txt"), "w").write(txt + "\n")

# HBM-side traffic per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) counts
# 128-B read requests at 64 B -> double it; WRITE_SIZE (KB) is exact for 16-B-per-lane streaming stores.
import json
cfg, run = {}, {}
try:
    run = json.load(open(os.path.join(out, "pass1.json")))
    cfg = run["config"]
except Exception:
    pass
steps_total = (run.get("steps", 0) + run.get("warmup", 0)) or None
traffic = {}
for k, cs in acc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        name = "trsm_chi2_kernel" if "trsm" in k else ("tri_gemm_chi2_kernel" if "tri_gemm" in k else ("walker_kernel" if "walker" in k else None))
        if name:
            per_step = 1  # one dispatch of each kernel per evaluation (the preconditioning evaluations are dispatches of the same shape)
            f = cs["FETCH_SIZE"][2:] or cs["FETCH_SIZE"]
            w = cs["WRITE_SIZE"][2:] or cs["WRITE_SIZE"]
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            traffic[name] = {"FETCH_SIZE_KB_per_dispatch": fk, "WRITE_SIZE_KB_per_dispatch": wk, "dispatches_per_step": per_step,
                             "hbm_bytes_per_launch": (2 * fk + wk) * 1024 * per_step}
json.dump({"config": {"n_sn": cfg.get("n_sn"), "walkers_per_gpu": cfg.get("walkers_per_gpu"), "n_grid": cfg.get("n_grid"),
                      "workload": cfg.get("workload_key", "pantheon")},
           "kernels": traffic,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python bench.py --steps 5 --warmup 2` "
                     "(preconditioned device); bytes per launch of W walkers = (2*FETCH_SIZE + WRITE_SIZE) KB per dispatch (gfx950 correction)"},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
