#!/bin/bash
# round 4, GPU call 17: the round's net effect on the headline kernel, same box, interleaved: round-3 library against this round's
# (and this round's with every tile multiplied)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_17; mkdir -p $O
for rep in 1 2 3 4 5; do
  for W in 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh n_w${W}_r3_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_r04base.so
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh n_w${W}_r4full_$rep CF_TUNE=gemm_diag_skip=0,gemm_trim=0
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh n_w${W}_r4_$rep
  done
done 2>&1 | tee $O/net_ab.txt
