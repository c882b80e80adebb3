#!/bin/bash
# round 3, GPU call 40: small_blocks_kernel with a wave per walker for batches of <= 256 walkers (width-independent sums)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_40; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_joint.py tests/test_variants.py tests/test_fs8.py tests/test_scripts.py tests/test_plot_accessors.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for wide in 0 256; do for wl in desi_cmb_des5y desi_cmb_des5y:cpl; do
  echo "== CF_SB_WIDE_MAX=$wide WORKLOAD=$wl"
  CF_SB_WIDE_MAX=$wide WORKLOAD=$wl WS=1,16,32,64,100,128,256 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done; done | tee $O/wall.txt
export BENCH_ARGS="--workload desi_cmb_des5y --fde cpl"; tools/quick_ab.sh c3cpl | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
WORKLOAD=desi_cmb_des5y:cpl WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace.log; exit 1; }
f=$(find $GRAFT_REPO_ROOT/$O/trace -name '*kernel_trace.csv' | head -1); python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 900 | tee $GRAFT_REPO_ROOT/$O/kernels.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace
