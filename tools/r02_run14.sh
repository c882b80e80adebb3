#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
P=$GRAFT_REPO_ROOT/cosmology-model-fit_amd
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'])"; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2 3; do
for v in "" _prev; do
  COSMOFIT_LIB=$P/libcosmofit_hip$v.so python3 bench.py --no-cpu-baseline --steps 200 > $O/bench$v.$rep.json 2>/dev/null; show $O/bench$v.$rep.json
done; done
for v in "" _prev; do
  COSMOFIT_LIB=$P/libcosmofit_hip$v.so python3 bench.py --no-cpu-baseline --workload desi_cmb_des5y --fde cpl > $O/bench_c3cpl$v.json 2>/dev/null; show $O/bench_c3cpl$v.json
done
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > /dev/null 2>&1
COSMOFIT_LIB=$P/libcosmofit_hip_stamps.so timeout -k 10 200 python tools/walker_stamps.py > $O/walker_stamps.txt 2>&1; cat $O/walker_stamps.txt
