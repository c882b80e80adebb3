#!/bin/bash
# round 3, GPU call 42: phases of small_blocks_kernel (in-kernel stamps), 16 and 4096 walkers, w0waCDM joint likelihood
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_42; mkdir -p $O
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so WS=16,4096 timeout -k 10 200 python tools/sb_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so CF_SB_WIDE_MAX=0 WS=16 timeout -k 10 200 python tools/sb_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
