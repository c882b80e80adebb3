#!/bin/bash
# round-2 GPU call 1: full GPU test-suite, baseline bench, chunk-overlap sweep, 2-rank self-launch rehearsal, rocprof stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json
for ch in "2048" "1024,3072" "1024,1024" "512,1792" "1024,1536" "2048,1024"; do
  CF_CHUNKS=$ch timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 > $O/bench_chunks_${ch/,/_}.json 2> $O/bench_chunks_${ch/,/_}.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_chunks_${ch/,/_}.json")); print("chunks $ch", d["value"], d["ms_per_step"], d["kernels_ms"])
except Exception as e: print("chunks $ch failed", e)
PY
done
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 > $O/bench_nochunk100.json 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_nochunk100.json')); print('nochunk', d['value'], d['ms_per_step'], d['kernels_ms'])"
BENCH_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --walkers-per-gpu 2048 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 400 $O/bench_gloo2.json
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 > $GRAFT_REPO_ROOT/$O/prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -8 $O/kernel_stats.csv
