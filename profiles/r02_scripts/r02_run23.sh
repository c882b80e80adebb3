#!/bin/bash
# walker_kernel phase stamps: Pantheon (late-time flat LCDM) vs the config-3 joint (physical densities), stamp build
cd $GRAFT_REPO_ROOT
O=gpurun_out/run23; mkdir -p $O
bash tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
V=$GRAFT_REPO_ROOT/cosmology-model-fit_amd/libcosmofit_hip_stamps.so
CF_ZEROCOPY_MAX=0 COSMOFIT_LIB=$V timeout -k 10 200 python tools/walker_stamps.py 2>/dev/null > $O/stamps_pantheon.txt; cat $O/stamps_pantheon.txt
CF_ZEROCOPY_MAX=0 WORKLOAD=desi_cmb_des5y COSMOFIT_LIB=$V timeout -k 10 200 python tools/walker_stamps.py 2>/dev/null > $O/stamps_config3.txt; cat $O/stamps_config3.txt
rm -f $V
