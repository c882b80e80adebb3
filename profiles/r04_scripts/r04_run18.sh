#!/bin/bash
# round 4, GPU call 18: guards only on the groups that need them: parity, then the net A/B against the round-3 library again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_18; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -60 $O/pytest.log; exit $rc; }
for rep in 1 2 3 4; do
  for W in 1024 2048 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh n_w${W}_r3_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_r04base.so
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh n_w${W}_r4full_$rep CF_TUNE=gemm_diag_skip=0,gemm_trim=0,gemm_split=0
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh n_w${W}_r4_$rep
  done
done 2>&1 | tee $O/net_ab.txt
