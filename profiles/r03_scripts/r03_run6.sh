#!/bin/bash
# round 3, GPU call 6: the lean production walker kernel: GPU tests, A/B against the generic kernel, stamps, bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_6; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for cfg in "pantheon:" "c3lcdm:--workload desi_cmb_des5y" "c3cpl:--workload desi_cmb_des5y --fde cpl" "c5:--workload desi_des5y_bbn_theta_star"; do
  export BENCH_ARGS="${cfg#*:}"
  tools/quick_ab.sh ${cfg%%:*}_lean
  tools/quick_ab.sh ${cfg%%:*}_generic CF_WALKER_GENERIC=1
done 2>&1 | tee $O/walker_lean_ab.txt
for fde in lcdm cpl; do echo "== config 3, fde = $fde"; WORKLOAD=desi FDE=$fde CF_ZEROCOPY_MAX=0 COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 300 python tools/walker_stamps.py || exit 1; done > $O/walker_stamps_config3.txt 2>&1
echo "== pantheon" >> $O/walker_stamps_config3.txt; CF_ZEROCOPY_MAX=0 COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 300 python tools/walker_stamps.py >> $O/walker_stamps_config3.txt 2>&1
grep -v amdgpu.ids $O/walker_stamps_config3.txt | grep "wg \|==" 
