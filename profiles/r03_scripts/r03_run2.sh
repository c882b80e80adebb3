#!/bin/bash
# round 3, GPU call 2: GPU tests with the small-batch solve kernel, its timeline, prefetch-depth and switch-point sweeps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for pf in 4 8 16; do echo "CF_SMALL_PF=$pf"; CF_SMALL_PF=$pf WS=1,16,32,64 timeout -k 10 300 python tools/small_batch_timeline.py || exit 1; done > $O/small_pf_sweep.txt 2>&1
cat $O/small_pf_sweep.txt
( echo "CF_SMALL_MAX=0 (throughput kernel)"; CF_SMALL_MAX=0 WS=16,64,128,256 timeout -k 10 300 python tools/small_batch_timeline.py &&
  echo "CF_SMALL_MAX=256"; CF_SMALL_MAX=256 WS=16,64,128,256 timeout -k 10 300 python tools/small_batch_timeline.py ) > $O/small_switch_sweep.txt 2>&1 || exit 1
cat $O/small_switch_sweep.txt
cd /tmp && export TMPDIR=/tmp &&
WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 &&
cd $GRAFT_REPO_ROOT && f=$(find $O/trace -name '*kernel_trace.csv' | head -1) && python tools/timeline_gaps.py $f 600 | tee $O/timeline_w16.txt
