#!/bin/bash
# round 3, GPU call 26: walker kernel duration at W = 16 with theta read in place from pinned host memory (zero-copy) against a device copy
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_26; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for zc in 4096 0; do
  CF_ZEROCOPY_MAX=$zc WS=1,16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$zc -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace$zc.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace$zc.log; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/$O/trace$zc -name '*kernel_trace.csv' | head -1)
  echo "== CF_ZEROCOPY_MAX=$zc"; grep "W=" $GRAFT_REPO_ROOT/$O/trace$zc.log; python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 600
done | tee $GRAFT_REPO_ROOT/$O/zc.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace*/
cd $GRAFT_REPO_ROOT
WS=1,16,32,64 REPS=400 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W=" | tee $O/wall.txt
