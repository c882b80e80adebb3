#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'])"; }
for rep in 1 2; do
for shape in 2x2 1x2 1x4 2x3 2x4; do
  CF_GEMM_SHAPE=$shape python3 bench.py --no-cpu-baseline --steps 200 > $O/bench_$shape.$rep.json 2>/dev/null; show $O/bench_$shape.$rep.json
done; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
