#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_3; mkdir -p $O
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for pf in 16 4; do echo "== CF_SMALL_PF=$pf"; CF_SMALL_PF=$pf COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 300 python tools/small_stamps.py || exit 1; done > $O/small_stamps.txt 2>&1
cat $O/small_stamps.txt | cut -c1-200
