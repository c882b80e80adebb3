#!/bin/bash
# round 3, GPU call 9: the last arriver's read-back with all loads in flight: GPU tests, small-batch timeline, latency probe, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_9; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/small_batch_timeline.py > $O/small_batch_wall.txt 2>&1 && grep -v amdgpu.ids $O/small_batch_wall.txt
timeout -k 10 300 python tools/latency_probe.py 2>&1 | grep "inverse" | tee $O/latency_probe.txt
tools/quick_ab.sh final_pantheon
cd /tmp && export TMPDIR=/tmp &&
WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 &&
cd $GRAFT_REPO_ROOT && f=$(find $O/trace -name '*kernel_trace.csv' | head -1) && python tools/timeline_gaps.py $f 600 | tee $O/timeline_w16.txt
