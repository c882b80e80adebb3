#!/bin/bash
# round 4, GPU call 16: half units for the lowest row blocks (CF_TUNE gemm_split=<levels>; 0 = none): parity, then levels x batch sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_16; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -60 $O/pytest.log; exit $rc; }
for rep in 1 2; do
  for W in 768 1024 2048 4096 8192; do
    for S in 0 4 8 12 16; do
      BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh h_w${W}_s${S}_$rep CF_TUNE=gemm_split=$S
    done
  done
done 2>&1 | tee $O/split_ab.txt
