#!/bin/bash
# round 3: completion word in pinned host memory for small synchronous calls: GPU tests, then A/B of the wall time per call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_18; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for f in 1 0; do echo "CF_DONE_FLAG=$f"; CF_DONE_FLAG=$f WS=1,16,32,48,75,150,512,2048 timeout -k 10 300 python tools/small_batch_timeline.py || exit 1; done; done 2>&1 | grep -v amdgpu.ids | tee $O/done_flag_ab.txt
