#!/bin/bash
# round 4, GPU call 9: the ceiling and the kernel on ONE box: a bare v_mfma_f64_16x16x4_f64 loop (registers only) at 1 / 2 / 4 waves per
# SIMD with its in-kernel clock, then the solve kernel's executed-MFMA rate and in-kernel clock (diagnostic build) and the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_9; mkdir -p $O
timeout -k 10 120 tools/coexec_f64_rate 2>&1 | tee $O/bare_mfma_rate.txt
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so timeout -k 10 300 python tools/solve_clock.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/solve_clock.txt
for rep in 1 2; do tools/quick_ab.sh prod_$rep; BENCH_ARGS="--walkers-per-gpu 8192" tools/quick_ab.sh prod8192_$rep; done 2>&1 | tee $O/bench.txt
timeout -k 10 120 tools/coexec_f64_rate 2>&1 | tee $O/bare_mfma_rate_again.txt
