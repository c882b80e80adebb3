#!/bin/bash
# round 3, GPU call 43: small_blocks_kernel reading theta across the lanes of a walker's group (ThetaGroup)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_43; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_joint.py tests/test_variants.py tests/test_fs8.py tests/test_scripts.py tests/test_plot_accessors.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so WS=16,4096 timeout -k 10 200 python tools/sb_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
for wl in desi_cmb_des5y desi_cmb_des5y:cpl; do
  echo "== WORKLOAD=$wl"
  WORKLOAD=$wl WS=1,16,64,100,256 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done | tee $O/wall.txt
for rep in 1 2; do for wl in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y"; do export BENCH_ARGS="--workload $wl"; tools/quick_ab.sh tg_$rep; done; done | tee $O/ab.txt
