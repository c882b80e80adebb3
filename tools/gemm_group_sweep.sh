#!/bin/bash
# effect of the panel-group size of the inverse-GEMM solve (CF_GEMM_GROUP = panels per group, 0 = one group)
mkdir -p gpurun_out
for grp in "$@"; do
  CF_GEMM_GROUP=$grp timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --solve inverse > gpurun_out/gg_$grp.json 2> gpurun_out/gg_$grp.err || tail -3 gpurun_out/gg_$grp.err
  python - <<PY
import json
d=json.load(open("gpurun_out/gg_$grp.json"))
print("[group $grp] evals/s=%.3e ms/step=%.3f walker=%.3f ms solve=%.3f ms (%.1f TF)"%(d["value"],d["ms_per_step"],d["kernels_ms"]["walker_kernel"],d["kernels_ms"][d["roofline"]["kernel"]],d["roofline"]["achieved"]))
PY
done
