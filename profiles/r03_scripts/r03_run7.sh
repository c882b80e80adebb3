#!/bin/bash
# round 3, GPU call 7: GPU tests (hand-off tests, growth block with the scripts' PCHIP), growth parity probe, zero-copy threshold probe,
# small-batch timeline with the lean walker kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_7; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -8 $O/pytest.log
timeout -k 10 300 python tools/fs8_parity_probe.py > $O/fs8_parity.txt 2>&1; grep -v amdgpu.ids $O/fs8_parity.txt
[ $rc -ne 0 ] && exit $rc
for zc in 0 65536; do CF_ZEROCOPY_MAX=$zc WS=2048,4096,8192,16384,32768 timeout -k 10 300 python tools/zerocopy_probe.py || exit 1; done > $O/zerocopy.txt 2>&1
grep -v amdgpu.ids $O/zerocopy.txt
timeout -k 10 300 python tools/small_batch_timeline.py > $O/small_batch_wall.txt 2>&1 && grep -v amdgpu.ids $O/small_batch_wall.txt
cd /tmp && export TMPDIR=/tmp &&
WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 &&
cd $GRAFT_REPO_ROOT && f=$(find $O/trace -name '*kernel_trace.csv' | head -1) && python tools/timeline_gaps.py $f 600 | tee $O/timeline_w16.txt
