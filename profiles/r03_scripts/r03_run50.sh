#!/bin/bash
# round 3, GPU call 50: order of the small-batch solve kernel's workgroups over several panels (CF_SMALL_ORDER 0: panel-major, 1: longest first + snake)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_50; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_gpu_random_shapes.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for o in 0 1; do
  echo "== CF_SMALL_ORDER=$o"
  CF_SMALL_ORDER=$o WS=16,32,48,64,75,96,100,128,150,160 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done; done | tee $O/wall.txt
