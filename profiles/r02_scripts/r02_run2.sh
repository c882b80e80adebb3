#!/bin/bash
# round-2 GPU call 2: hand-off A/B (relaxed vs release add), traffic-alias ceilings, two-engine overlap probe, rocprof csv stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/cosmology-model-fit_amd
for rep in 1 2; do
for v in "" _rel _aliasA _aliasAB; do
  COSMOFIT_LIB=$P/libcosmofit_hip$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 > $O/bench$v.$rep.json 2> $O/bench$v.$rep.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench$v.$rep.json")); print("variant '$v' rep $rep", "%.4e" % d["value"], "%.4f" % d["ms_per_step"], d["kernels_ms"])
except Exception as e: print("variant $v failed", e)
PY
done
done
timeout -k 10 300 python tools/overlap_probe.py 4096 > $O/overlap_probe.txt 2>&1; cat $O/overlap_probe.txt
timeout -k 10 300 python tools/overlap_probe.py 2048 >> $O/overlap_probe.txt 2>&1; tail -4 $O/overlap_probe.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 > $GRAFT_REPO_ROOT/$O/prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -8 $O/kernel_stats.csv
