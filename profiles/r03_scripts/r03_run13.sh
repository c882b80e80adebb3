#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_13; mkdir -p $O
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 300 python tools/gemm_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_stamps.txt
