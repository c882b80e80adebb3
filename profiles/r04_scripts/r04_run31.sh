#!/bin/bash
# round 4, GPU call 31: small-blocks kernel with 32 lanes per walker at 4096 walkers (variant build; production: 16), joint workloads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_31; mkdir -p $O
L=$PWD/cosmology-model-fit_amd/libcosmofit_hip_sb32.so
for rep in 1 2 3; do
  for cfg in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y" "desi_des5y_bbn_theta_star"; do
    tag=$(echo $cfg | tr ' -' '__')
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh sb_${tag}_lanes16_$rep COSMOFIT_LIB=$L
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh sb_${tag}_lanes32_$rep COSMOFIT_LIB=$L CF_TUNE=sb_narrow_lanes=32
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh sb_${tag}_lanes64_$rep COSMOFIT_LIB=$L CF_TUNE=sb_wide_max=100000,sb_roles_max=0
  done
done 2>&1 | tee $O/sb_lanes.txt
