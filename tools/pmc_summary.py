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
    if not ("trsm" in k or "sn_residual" in k):
        continue
    lines.append(f"== {k[:60]}  (dispatches: {max(len(v) for v in cs.values())})")
    for c, v in sorted(cs.items()):
        v = v[2:] if len(v) > 3 else v  # drop warm-up dispatches
        lines.append(f"   {c:36s} {sum(v)/len(v):18.1f}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "pmc_summary.txt"), "w").write(txt + "\n")
