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
    if not ("trsm" in k or "tri_gemm" in k or "walker" in k):
        continue
    lines.append(f"== {k[:60]}  (dispatches: {max(len(v) for v in cs.values())})")
    for c, v in sorted(cs.items()):
        v = v[2:] if len(v) > 3 else v  # drop warm-up dispatches
        lines.append(f"   {c:36s} {sum(v)/len(v):18.1f}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "pmc_summary.txt"), "w").write(txt + "\n")

# HBM-side traffic per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) counts
# 128-B read requests at 64 B -> double it; WRITE_SIZE (KB) is exact for 16-B-per-lane streaming stores.
import json
traffic = {}
for k, cs in acc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        name = "trsm_chi2_kernel" if "trsm" in k else ("tri_gemm_chi2_kernel" if "tri_gemm" in k else ("walker_kernel" if "walker" in k else None))
        if name:
            f = cs["FETCH_SIZE"][2:] or cs["FETCH_SIZE"]
            w = cs["WRITE_SIZE"][2:] or cs["WRITE_SIZE"]
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            traffic[name] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
cfg = {}
try:
    cfg = json.load(open(os.path.join(out, "pass1.json")))["config"]
except Exception:
    pass
json.dump({"config": {k: cfg.get(k) for k in ("n_sn", "walkers_per_gpu", "n_grid")}, "kernels": traffic,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction)"},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
