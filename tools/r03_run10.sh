#!/bin/bash
# round 3, GPU call 10: the fused small-batch kernel: parity tests first (bounded), then A/B against the two-launch path, then the whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_10; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_invariance or config2 or golden or nonfinite or device_ensemble" > $O/pytest_first.log 2>&1; rc=$?; tail -5 $O/pytest_first.log
[ $rc -ne 0 ] && exit $rc
for f in 1 0; do echo "CF_FUSED=$f"; CF_FUSED=$f timeout -k 10 300 python tools/small_batch_timeline.py || exit 1; done > $O/fused_ab.txt 2>&1
grep -v amdgpu.ids $O/fused_ab.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp &&
WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 &&
cd $GRAFT_REPO_ROOT && f=$(find $O/trace -name '*kernel_trace.csv' | head -1) && python tools/timeline_gaps.py $f 300 | tee $O/timeline_w16.txt
