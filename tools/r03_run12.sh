#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_12; mkdir -p $O
tools/build_variant.sh nofence -DCF_TRSM_STAMPS -DCF_DBG_FUSED_NOFENCE > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_nofence.so timeout -k 10 300 python tools/fused_stamps.py > $O/fused_stamps_nofence.txt 2>&1; grep -v amdgpu.ids $O/fused_stamps_nofence.txt | sed -n 18,30p
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_nofence.so SKIP_CHECK=1 WS=16 timeout -k 10 300 python tools/small_batch_timeline.py 2>&1 | grep "W="
