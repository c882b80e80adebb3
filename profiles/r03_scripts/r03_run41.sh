#!/bin/bash
# round 3, GPU call 41: up to which batch size does a wave per walker pay in small_blocks_kernel?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_41; mkdir -p $O
for rep in 1 2; do for wide in 0 4096; do
  echo "== CF_SB_WIDE_MAX=$wide WORKLOAD=desi_cmb_des5y:cpl"
  CF_SB_WIDE_MAX=$wide WORKLOAD=desi_cmb_des5y:cpl WS=256,512,1024,2048,4096 REPS=200 SKIP_CHECK=1 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done; done | tee $O/wall.txt
