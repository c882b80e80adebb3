#!/bin/bash
# round 4, GPU call 5: in-kernel timeline of the work-queue solve kernel (diagnostic build): where do the 3-7 us against the static grid go?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_5; mkdir -p $O
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so timeout -k 10 300 python tools/solve_clock.py 1024 2048 4096 2>&1 | grep -v amdgpu.ids | tee $O/solve_clock.txt
