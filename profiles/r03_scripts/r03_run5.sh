#!/bin/bash
# round 3, GPU call 5: A/B of the table-driven exp (CPL table build) on one box + walker phase stamps for the CPL joint config
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_scripts.py tests/test_gpu_sharded_native.py -m gpu -x -q 2>&1 | tail -3
tools/build_variant.sh libexp -DCF_LIB_EXP > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
tools/build_variant.sh stamps -DCF_TRSM_STAMPS >> $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
export BENCH_ARGS="--workload desi_cmb_des5y --fde cpl"
for rep in 1 2; do
  tools/quick_ab.sh exp_tab_$rep
  tools/quick_ab.sh lib_exp_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_libexp.so
done 2>&1 | tee $O/cpl_exp_ab.txt
for fde in lcdm cpl; do echo "== config 3, fde = $fde"; WORKLOAD=desi FDE=$fde CF_ZEROCOPY_MAX=0 COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 300 python tools/walker_stamps.py || exit 1; done > $O/walker_stamps_config3.txt 2>&1
cat $O/walker_stamps_config3.txt
