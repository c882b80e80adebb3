#!/bin/bash
# round 3, GPU call 51: panel width of the throughput solve kernel for 256-1024 walkers now that small grids run in snake order
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_51; mkdir -p $O
for rep in 1 2; do for sh in 1x2 2x2; do
  echo "== CF_GEMM_SHAPE=$sh"
  CF_DONE_FLAG=1 CF_GEMM_SHAPE=$sh WS=256,384,512,640,768,896,1024 REPS=200 SKIP_CHECK=1 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done; done | tee $O/wall.txt
