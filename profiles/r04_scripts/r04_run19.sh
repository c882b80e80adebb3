#!/bin/bash
# round 4, GPU call 19: half units (12 levels) on / off over the mid sizes with the final kernel (guards only where needed)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_19; mkdir -p $O
for rep in 1 2 3; do
  for W in 640 768 1024 1536 2048 3072; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh s_w${W}_r3_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_r04base.so
    for S in 0 6 12; do BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh s_w${W}_s${S}_$rep CF_TUNE=gemm_split=$S; done
  done
done 2>&1 | tee $O/split_ab.txt
