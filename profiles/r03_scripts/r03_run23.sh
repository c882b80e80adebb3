#!/bin/bash
# round 3, GPU call 23: residuals of a small batch in the solve kernel's fragment order (CF_SMALL_FRAG=1, default) against walker rows (0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_23; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_variants.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for fr in 0 1; do for pf in 4 8 16; do
  echo "== CF_SMALL_FRAG=$fr CF_SMALL_PF=$pf"
  CF_SMALL_FRAG=$fr CF_SMALL_PF=$pf WS=1,16,32,64 REPS=400 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done; done | tee $O/wall.txt
cd /tmp && export TMPDIR=/tmp
for fr in 0 1; do for pf in 4 16; do
  CF_SMALL_FRAG=$fr CF_SMALL_PF=$pf WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$fr$pf -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace$fr$pf.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace$fr$pf.log; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/$O/trace$fr$pf -name '*kernel_trace.csv' | head -1)
  echo "== CF_SMALL_FRAG=$fr CF_SMALL_PF=$pf"; python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 600
done; done | tee $GRAFT_REPO_ROOT/$O/kernels.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace*/
