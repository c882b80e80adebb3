#!/usr/bin/env python3
"""Per-kernel mean duration and the idle gap in front of each kernel from a rocprofv3 --kernel-trace CSV
(columns Kernel_Name, Start_Timestamp, End_Timestamp): python tools/timeline_gaps.py <kernel_trace.csv> [last N dispatches]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None:
        gap[name].append(s - prev_end)
    prev_end = e
print(f"{'kernel':60s} {'n':>6s} {'mean us':>8s} {'p50 us':>8s} {'gap before: mean':>16s} {'p50':>8s}")
for k in dur:
    d, g = sorted(dur[k]), sorted(gap[k]) or [0]
    print(f"{k:60s} {len(d):6d} {sum(d) / len(d) / 1e3:8.2f} {d[len(d) // 2] / 1e3:8.2f} {sum(g) / len(g) / 1e3:16.2f} {g[len(g) // 2] / 1e3:8.2f}")
