#!/bin/bash
# A/B of the solve kernels on the GPU box: blocked TRSM vs the inverse-GEMM shapes (CF_GEMM_SHAPE=<NP>x<PF>).
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/gs_$tag.json 2> gpurun_out/gs_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/gs_$tag.err; }
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/gs_$tag.json"))
    print("[$tag] evals/s=%.3e ms/step=%.3f walker=%.3f ms solve=%.3f ms (%.1f TF)"%(d["value"],d["ms_per_step"],d["kernels_ms"]["walker_kernel"],d["kernels_ms"][d["roofline"]["kernel"]],d["roofline"]["achieved"]))
except Exception as e: print("[$tag] no result", e)
PY
}
BENCH_ARGS="--solve blocked" run blocked A=1
for shape in "$@"; do BENCH_ARGS="--solve inverse" run inv_$shape CF_GEMM_SHAPE=$shape; done
