#!/bin/bash
# round 3, GPU call 32: 2 / 4 tiles per workgroup of the small-batch solve kernel beyond 128 walkers, against the throughput kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_32; mkdir -p $O
for rep in 1 2; do
  echo "== throughput kernel (CF_SMALL_MAX=0)"; CF_SMALL_MAX=0 WS=128,150,200,256 REPS=400 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
  for t in 2 4; do for pf in 4 8; do
    [ $t = 4 ] && [ $pf = 8 ] && continue
    echo "== CF_SMALL_MAX=256 CF_SMALL_TPW=$t CF_SMALL_PF=$pf"
    CF_SMALL_MAX=256 CF_SMALL_TPW=$t CF_SMALL_PF=$pf WS=100,128,150,200,256 REPS=400 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
  done; done
done | tee $O/wall.txt
