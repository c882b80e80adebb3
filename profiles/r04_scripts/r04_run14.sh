#!/bin/bash
# round 4, GPU call 14: zero tiles of the diagonal blocks not multiplied (wave 3's last four groups): parity, A/B against multiplying them
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_14; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -60 $O/pytest.log; exit $rc; }
for rep in 1 2 3; do
  for W in 512 1024 2048 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh d_w${W}_full_$rep CF_TUNE=gemm_diag_skip=0,gemm_trim=0
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh d_w${W}_trim_$rep CF_TUNE=gemm_diag_skip=0
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh d_w${W}_skip_$rep
  done
done 2>&1 | tee $O/diag_skip_ab.txt
