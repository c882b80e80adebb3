#!/bin/bash
# round 3, GPU call 49: snake order as shipped (grids of <= 1024 workgroups): tests, wall times, the 4096-walker bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_49; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for o in 0 -1; do
  echo "== CF_GEMM_ORDER=$o (-1: default rule)"
  if [ $o = -1 ]; then WS=161,256,384,512,640,768,1024,1536 REPS=200 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="; else CF_GEMM_ORDER=$o WS=161,256,384,512,640,768,1024,1536 REPS=200 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="; fi
done; done | tee $O/wall.txt
unset BENCH_ARGS; for rep in 1 2; do tools/quick_ab.sh final_$rep; done | tee $O/ab.txt
