#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
P=$GRAFT_REPO_ROOT/cosmology-model-fit_amd
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'])"; }
for rep in 1 2 3; do
for v in "" _snv1; do
  COSMOFIT_LIB=$P/libcosmofit_hip$v.so python3 bench.py --no-cpu-baseline --steps 200 > $O/bench$v.$rep.json 2>/dev/null; show $O/bench$v.$rep.json
done; done
for v in "" _snv1; do
  COSMOFIT_LIB=$P/libcosmofit_hip$v.so python3 bench.py --no-cpu-baseline --workload desi_cmb_des5y > $O/bench_c3$v.json 2>/dev/null; show $O/bench_c3$v.json
done
timeout -k 10 300 python tools/ensemble_probe.py > $O/ensemble_probe.txt 2>&1; cat $O/ensemble_probe.txt
