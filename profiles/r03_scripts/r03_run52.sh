#!/bin/bash
# round 3, GPU call 52: 32-walker panels from 577 walkers (was 769): default path with completion words
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_52; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "invariance or soak or configs3" > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do WS=512,576,577,640,704,768,769,896,1024 REPS=200 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="; done | tee $O/wall.txt
