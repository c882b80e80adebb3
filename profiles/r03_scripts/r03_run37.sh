#!/bin/bash
# round 3, GPU call 37: lanes per walker in small_blocks_kernel (16 as shipped, 32, 64): one wave per SIMD at 16 lanes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_37; mkdir -p $O
for l in 32 64; do tools/build_variant.sh sb$l -DCF_SB_LANES=$l > $O/build$l.log 2>&1 || { tail $O/build$l.log; exit 1; }; done
for rep in 1 2; do
  for wl in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y" "desi_des5y_bbn_theta_star"; do
    export BENCH_ARGS="--workload $wl"
    echo "== $wl"
    tools/quick_ab.sh l16_$rep
    tools/quick_ab.sh l32_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_sb32.so
    tools/quick_ab.sh l64_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_sb64.so
  done
done 2>&1 | tee $O/ab.txt
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_sb64.so timeout -k 10 600 python -m pytest tests/test_gpu_joint.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest64.txt
