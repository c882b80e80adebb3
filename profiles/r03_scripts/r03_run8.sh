#!/bin/bash
# round 3, GPU call 8: where the 20 us of the small-batch solve kernel go: truncated debug builds under rocprofv3 --kernel-trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_8; mkdir -p $O
for v in 1 2 3; do tools/build_variant.sh dbg$v -DCF_DBG_SMALL=$v > $O/build$v.log 2>&1 || { tail $O/build$v.log; exit 1; }; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  lib=$GRAFT_REPO_ROOT/cosmology-model-fit_amd/libcosmofit_hip_dbg$v.so; [ $v = 0 ] && lib=$GRAFT_REPO_ROOT/cosmology-model-fit_amd/libcosmofit_hip.so
  COSMOFIT_LIB=$lib SKIP_CHECK=1 WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$v -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace$v.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace$v.log; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/$O/trace$v -name '*kernel_trace.csv' | head -1)
  echo "== CF_DBG_SMALL=$v"; grep "W=" $GRAFT_REPO_ROOT/$O/trace$v.log; python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 600
done | tee $GRAFT_REPO_ROOT/$O/small_kernel_breakdown.txt
