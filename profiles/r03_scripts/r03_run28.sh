#!/bin/bash
# round 3, GPU call 28: what would contiguous residual fragments be worth to the THROUGHPUT solve kernel? (timing build, wrong results)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_28; mkdir -p $O
tools/build_variant.sh fragb -DCF_DEBUG_FRAG_B > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
for rep in 1 2 3; do
  tools/quick_ab.sh base_$rep
  tools/quick_ab.sh fragb_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_fragb.so
done 2>&1 | tee $O/ab.txt
