#!/bin/bash
# round 3, GPU call 1: the GPU test-suite (incl. the sharded native-kernel test), then the small-batch timeline of the round-2 path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/small_batch_timeline.py > $O/small_batch_wall.txt 2>&1 && cat $O/small_batch_wall.txt &&
cd /tmp && export TMPDIR=/tmp &&
WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 &&
cd $GRAFT_REPO_ROOT && f=$(find $O/trace -name '*kernel_trace.csv' | head -1) && python tools/timeline_gaps.py $f 600 | tee $O/timeline_w16.txt
