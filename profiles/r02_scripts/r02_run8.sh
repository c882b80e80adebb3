#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'], d['roofline']['traffic_source'], d.get('value_host_visible'))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; show $O/bench_driver_cmd.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --precondition-ms 0 --no-cpu-baseline > $O/bench_cold.json 2>/dev/null; show $O/bench_cold.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null; show $O/bench_default.json
CF_GEMM_GROUP=0 python3 bench.py --scaling strong --no-cpu-baseline --steps 10 > $O/bench_strong_nogroup.json 2>/dev/null; show $O/bench_strong_nogroup.json
python3 bench.py --scaling strong --no-cpu-baseline --steps 10 > $O/bench_strong_group256.json 2>/dev/null; show $O/bench_strong_group256.json
CF_GEMM_GROUP=128 python3 bench.py --scaling strong --no-cpu-baseline --steps 10 > $O/bench_strong_group128.json 2>/dev/null; show $O/bench_strong_group128.json
python3 bench.py --walkers-per-gpu 16384 --no-cpu-baseline --steps 20 > $O/bench_w16384.json 2>/dev/null; show $O/bench_w16384.json
CF_GEMM_GROUP=0 python3 bench.py --walkers-per-gpu 16384 --no-cpu-baseline --steps 20 > $O/bench_w16384_nogroup.json 2>/dev/null; show $O/bench_w16384_nogroup.json
BENCH_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
