#!/bin/bash
# small blocks / growth kernel beside the solve: bit-identity test, joint + fs8 tests, then the joint benches one stream vs two
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/run22; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_joint.py tests/test_fs8.py tests/test_scripts.py tests/test_variants.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
show() { python -c "
import json; d=json.load(open('$1')); print('$2', '%.4e'%d['value'], '%.4f ms'%d['ms_per_step'], d['kernels_ms'])"; }
for wl in desi_cmb_des5y desi_des5y_bbn_theta_star; do
  for rep in 1 2; do
    CF_OVERLAP_SMALL=0 python3 bench.py --workload $wl --no-cpu-baseline > $O/${wl}_one_$rep.json 2>/dev/null; show $O/${wl}_one_$rep.json "one stream  $wl"
    CF_OVERLAP_SMALL=1 python3 bench.py --workload $wl --no-cpu-baseline > $O/${wl}_two_$rep.json 2>/dev/null; show $O/${wl}_two_$rep.json "beside      $wl"
  done
done
CF_OVERLAP_SMALL=0 python3 bench.py --workload desi_cmb_des5y --fde cpl --no-cpu-baseline > $O/cpl_one.json 2>/dev/null; show $O/cpl_one.json "one stream  cpl"
CF_OVERLAP_SMALL=1 python3 bench.py --workload desi_cmb_des5y --fde cpl --no-cpu-baseline > $O/cpl_two.json 2>/dev/null; show $O/cpl_two.json "beside      cpl"
echo "== fs8 probe, one stream"; CF_OVERLAP_SMALL=0 timeout -k 10 200 python tools/fs8_probe.py 2>/dev/null
echo "== fs8 probe, beside";     CF_OVERLAP_SMALL=1 timeout -k 10 200 python tools/fs8_probe.py 2>/dev/null
