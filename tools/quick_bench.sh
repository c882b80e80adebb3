#!/bin/bash
# usage: tools/quick_bench.sh tag [full] [env assignments...] -- runs GPU tests (subset unless "full") and a short bench
tag=$1; shift
sel='-k solve_triangular_vs_oracle or config2_full or golden'
if [ "$1" = "full" ]; then sel=''; shift; fi
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
if [ -n "$sel" ]; then
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "solve_triangular_vs_oracle or config2_full or golden" > gpurun_out/qt_$tag.log 2>&1
else
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/qt_$tag.log 2>&1
fi
rc=$?
echo "[$tag] pytest exit $rc : $(tail -1 gpurun_out/qt_$tag.log)"
if [ $rc -ne 0 ]; then tail -40 gpurun_out/qt_$tag.log; fi
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/qb_$tag.json 2> gpurun_out/qb_$tag.err || tail -3 gpurun_out/qb_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/qb_$tag.json"))
print("[$tag] evals/s=%.3e ms/step=%.3f resid=%.3f ms solve=%.3f ms (%.1f TF, %.1f%%)"%(d["value"],d["ms_per_step"],d["kernels_ms"]["walker_kernel"],d["kernels_ms"][d["roofline"]["kernel"]],d["roofline"]["achieved"],100*d["roofline"]["frac"]))
PY
