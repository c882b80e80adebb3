#!/bin/bash
# round 3, GPU call 38: small_blocks_kernel with two Gauss-Legendre nodes per iteration (interleaved chains)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_38; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_joint.py tests/test_variants.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for wl in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y" "desi_des5y_bbn_theta_star"; do
    export BENCH_ARGS="--workload $wl"; echo "== $wl"; tools/quick_ab.sh gl2_$rep
  done
done 2>&1 | tee $O/ab.txt
