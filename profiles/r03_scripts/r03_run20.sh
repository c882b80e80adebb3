#!/bin/bash
# round 3: where should the small-batch solve kernel hand over to the throughput kernel now (epilogue fixed, completion word)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_20; mkdir -p $O
for rep in 1 2; do for m in 0 256; do echo "CF_SMALL_MAX=$m"; CF_SMALL_MAX=$m WS=32,48,64,75,96,128,150,256 timeout -k 10 300 python tools/small_batch_timeline.py || exit 1; done; done 2>&1 | grep -v amdgpu.ids | tee $O/small_switch.txt
