#!/bin/bash
# round 3, GPU call 31: tiles per workgroup of the small-batch solve kernel (CF_SMALL_TPW = 1, 2, 4) for 32-128 walkers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_31; mkdir -p $O
for t in 2 4; do
  CF_SMALL_TPW=$t timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py -m gpu -x -q -k "invariance or small or native or config3_full" > $O/pytest$t.log 2>&1; rc=$?; tail -2 $O/pytest$t.log
  [ $rc -ne 0 ] && exit $rc
done
for rep in 1 2; do for t in 1 2 4; do
  echo "== CF_SMALL_TPW=$t"
  CF_SMALL_TPW=$t WS=16,32,48,64,75,96,100,128 REPS=400 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done; done | tee $O/wall.txt
