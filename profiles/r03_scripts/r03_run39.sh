#!/bin/bash
# round 3, GPU call 39: small batches of the JOINT likelihood (configs[2] shape): wall time per call and the three kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_39; mkdir -p $O
for wl in desi_cmb_des5y desi_cmb_des5y:cpl; do
  echo "== WORKLOAD=$wl"
  WORKLOAD=$wl WS=1,16,32,64,100,128 REPS=300 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done | tee $O/wall.txt
cd /tmp && export TMPDIR=/tmp
WORKLOAD=desi_cmb_des5y:cpl WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace.log; exit 1; }
f=$(find $GRAFT_REPO_ROOT/$O/trace -name '*kernel_trace.csv' | head -1); python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 900 | tee $GRAFT_REPO_ROOT/$O/kernels.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace
