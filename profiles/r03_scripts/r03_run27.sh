#!/bin/bash
# round 3, GPU call 27: several workgroups per walker in the per-walker kernel of a small batch (CF_SN_PARTS)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_27; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_variants.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for sp in 1 2 4 0; do
  echo "== CF_SN_PARTS=$sp (0: default)"
  CF_SN_PARTS=$sp WS=1,16,32,48,64 REPS=400 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done | tee $O/wall.txt
cd /tmp && export TMPDIR=/tmp
for sp in 1 4; do
  CF_SN_PARTS=$sp WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$sp -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace$sp.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace$sp.log; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/$O/trace$sp -name '*kernel_trace.csv' | head -1)
  echo "== CF_SN_PARTS=$sp"; python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 600
done | tee $GRAFT_REPO_ROOT/$O/kernels.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace*/
