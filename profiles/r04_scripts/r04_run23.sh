#!/bin/bash
# round 4, GPU call 23: TIMING EXPERIMENT (variant build, never shipped): the small-blocks kernel on a second stream BESIDE the walker
# kernel (it reads the previous evaluation's table nodes: same theta every step in the bench, so even the numbers are right) -- the upper bound of what
# splitting it into a theta-only part and a BAO part could gain for the joint configurations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_23; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y" "desi_des5y_bbn_theta_star"; do
    tag=$(echo $cfg | tr ' -' '__')
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh o_w4096_${tag}_seq_$rep
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh o_w4096_${tag}_overlap_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_sboverlap.so
  done
done 2>&1 | tee $O/sb_overlap.txt
