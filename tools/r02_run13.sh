#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
COSMOFIT_LIB=$GRAFT_REPO_ROOT/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 200 python tools/walker_stamps.py > $O/walker_stamps.txt 2>&1; cat $O/walker_stamps.txt
