#!/bin/bash
# round 4, GPU call 12: the last row block without its all-padding tiles (CF_TUNE gemm_trim=0 computes them): parity, A/B on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_12; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
for rep in 1 2 3; do
  for W in 1024 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh t_w${W}_full_$rep CF_TUNE=gemm_trim=0
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh t_w${W}_trim_$rep
  done
  BENCH_ARGS="--workload desi_cmb_des5y --fde cpl" tools/quick_ab.sh c3cpl_w4096_full_$rep CF_TUNE=gemm_trim=0
  BENCH_ARGS="--workload desi_cmb_des5y --fde cpl" tools/quick_ab.sh c3cpl_w4096_trim_$rep
done 2>&1 | tee $O/trim_ab.txt
