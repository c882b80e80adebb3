#!/bin/bash
# round 3, GPU call 30: the throughput solve kernel's last-arriver epilogue from LDS copies (descriptor, theta rows), straight-line share loads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_30; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_variants.py tests/test_gpu_handoff.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
WS=129,150,256,512,1024,2048 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W=" | tee $O/wall.txt
for rep in 1 2 3; do tools/quick_ab.sh new_$rep; done 2>&1 | tee $O/ab.txt
