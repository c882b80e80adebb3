#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
python3 bench.py --workload desi_cmb_des5y --fde cpl --no-cpu-baseline > gpurun_out/final/bench_config3_cpl.json 2>/dev/null
python3 bench.py --workload desi_cmb_des5y --no-cpu-baseline > gpurun_out/final/bench_config3_lcdm.json 2>/dev/null
python3 bench.py --workload desi_des5y_bbn_theta_star --no-cpu-baseline > gpurun_out/final/bench_config5.json 2>/dev/null
for f in config3_cpl config3_lcdm config5; do python - <<PY
import json; d=json.load(open("gpurun_out/final/bench_$f.json")); print("$f", "%.4e" % d["value"], "%.4f" % d["ms_per_step"], d["kernels_ms"])
PY
done
