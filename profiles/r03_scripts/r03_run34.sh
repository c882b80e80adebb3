#!/bin/bash
# round 3, GPU call 34: the small-batch path as configured (one tile per workgroup up to 96 walkers, two up to 160, two workgroups per CU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_34; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_variants.py tests/test_gpu_handoff.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  echo "== default"; WS=1,16,32,48,50,64,75,96,100,128,150,160,161,200 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
  echo "== CF_SMALL_LDS_CAP=0"; CF_SMALL_LDS_CAP=0 WS=16,48,64,75,100,128,150 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done | tee $O/wall.txt
