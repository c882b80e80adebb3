#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'])"; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python3 bench.py --workload desi_cmb_des5y --fde cpl > $O/bench_config3_cpl.json 2>/dev/null; show $O/bench_config3_cpl.json
python3 bench.py --workload desi_cmb_des5y > $O/bench_config3_lcdm.json 2>/dev/null; show $O/bench_config3_lcdm.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>/dev/null; show $O/bench.json
