#!/bin/bash
# round 3, GPU call 29: workgroups per walker for 65-128 walkers (small-batch solve kernel up to 128 walkers)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_29; mkdir -p $O
for rep in 1 2; do for sp in 1 2 4; do echo "CF_SN_PARTS=$sp"; CF_SN_PARTS=$sp WS=64,75,96,128 timeout -k 10 300 python tools/small_batch_timeline.py || exit 1; done; done 2>&1 | grep -v amdgpu.ids | tee $O/parts.txt
