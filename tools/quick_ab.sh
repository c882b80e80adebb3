#!/bin/bash
# usage: tools/quick_ab.sh tag [ENV=val ...] -- short bench (default solve) with optional env, one summary line
tag=$1; shift
mkdir -p gpurun_out
env "$@" A=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/ab_$tag.err; }
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_$tag.json"))
    print("[$tag] evals/s=%.3e ms/step=%.3f walker=%.3f ms small blocks=%.3f ms solve=%.3f ms (%.1f TF, %.3f)"%(d["value"],d["ms_per_step"],d["kernels_ms"]["walker_kernel"],d["kernels_ms"].get("small_blocks_kernel") or 0.0,d["kernels_ms"][d["roofline"]["kernel"]],d["roofline"]["achieved"],d["roofline"]["frac"]))
except Exception as e: print("[$tag] no result", e)
PY
