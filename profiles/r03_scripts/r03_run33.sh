#!/bin/bash
# round 3, GPU call 33: does capping the small-batch solve workgroups per CU (through LDS padding) spread them better?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_33; mkdir -p $O
for rep in 1 2; do
  for cfg in "1 0" "1 12000" "1 25000" "2 0" "2 4000" "2 17000"; do
    set -- $cfg
    echo "== CF_SMALL_TPW=$1 CF_SMALL_LDS_PAD=$2"
    CF_SMALL_MAX=256 CF_SMALL_TPW=$1 CF_SMALL_LDS_PAD=$2 WS=48,64,75,100,128,150,200 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
  done
done | tee $O/wall.txt
