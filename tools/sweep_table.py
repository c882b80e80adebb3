#!/usr/bin/env python3
"""Table of tools/quick_ab.sh lines tagged <prefix>_w<W>_<variant>_<rep>: solve-kernel fraction of the FP64 matrix peak per batch size and variant."""
import collections, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
variants = []
for l in open(sys.argv[1]):
    m = re.match(r'\[\w+?_w(\d+)_(\w+)_(\d)\].*solve=([\d.]+) ms \(([\d.]+) TF, ([\d.]+)\)', l)
    if m:
        W, v, rep, ms, tf, fr = m.groups()
        d[int(W)][v].append((float(fr), float(ms)))
        if v not in variants: variants.append(v)
print("W     " + "".join("%-22s" % v for v in variants))
for W in sorted(d):
    print("%-6d" % W + "".join("%-22s" % ("/".join("%.3f" % x[0] for x in d[W][v]) + " (%.0f us)" % (1e3 * min(x[1] for x in d[W][v]))) for v in variants))
