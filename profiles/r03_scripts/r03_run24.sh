#!/bin/bash
# round 3, GPU call 24: in-kernel stamps of the small-batch solve kernel with fragment-ordered residuals, incl. the last arriver's epilogue
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_24; mkdir -p $O
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 300 python tools/small_stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps.txt; head -12 $O/stamps.txt; tail -8 $O/stamps.txt
