#!/bin/bash
# round 3, GPU call 45: small_blocks_kernel with two waves per walker (BAO / CC beside powers + CMB integrals) for small batches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_45; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_joint.py tests/test_variants.py tests/test_fs8.py tests/test_scripts.py tests/test_plot_accessors.py tests/test_gpu_random_shapes.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for rm in 0 512 2048; do for wl in desi_cmb_des5y:cpl; do
  echo "== CF_SB_ROLES_MAX=$rm WORKLOAD=$wl"
  CF_SB_ROLES_MAX=$rm WORKLOAD=$wl WS=1,16,64,100,256,512,1024,2048 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done; done; done | tee $O/wall.txt
cd /tmp && export TMPDIR=/tmp
WORKLOAD=desi_cmb_des5y:cpl WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace.log; exit 1; }
f=$(find $GRAFT_REPO_ROOT/$O/trace -name '*kernel_trace.csv' | head -1); python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 900 | tee $GRAFT_REPO_ROOT/$O/kernels.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace
