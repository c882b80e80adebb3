#!/bin/bash
# round 3, GPU call 22: is the small-batch solve kernel's K loop bound by its loads?  (a dependent FP64 MFMA chain issues at 64 cycles per
# MFMA: tools/mfma_chain_latency.hip)  Builds whose factor / residual loads alias a few KiB, loads + MFMA loop only (CF_DBG_SMALL=2).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_22; mkdir -p $O
tools/build_variant.sh s2 -DCF_DBG_SMALL=2 > $O/b0.log 2>&1 || { tail $O/b0.log; exit 1; }
tools/build_variant.sh s2a -DCF_DBG_SMALL=2 -DCF_SMALL_ALIAS_A > $O/b1.log 2>&1 || { tail $O/b1.log; exit 1; }
tools/build_variant.sh s2b -DCF_DBG_SMALL=2 -DCF_SMALL_ALIAS_B > $O/b2.log 2>&1 || { tail $O/b2.log; exit 1; }
tools/build_variant.sh s2ab -DCF_DBG_SMALL=2 -DCF_SMALL_ALIAS_A -DCF_SMALL_ALIAS_B > $O/b3.log 2>&1 || { tail $O/b3.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for v in s2 s2a s2b s2ab; do
  for pf in 16 4; do
  lib=$GRAFT_REPO_ROOT/cosmology-model-fit_amd/libcosmofit_hip_$v.so
  CF_SMALL_PF=$pf COSMOFIT_LIB=$lib SKIP_CHECK=1 WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$v$pf -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace$v$pf.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace$v$pf.log; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/$O/trace$v$pf -name '*kernel_trace.csv' | head -1)
  echo "== variant $v PF=$pf"; python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 600 | grep small
  done
done | tee $GRAFT_REPO_ROOT/$O/alias.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace*/
