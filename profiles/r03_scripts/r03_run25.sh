#!/bin/bash
# round 3, GPU call 25: small-batch solve kernel, last arriver's epilogue reading descriptor / theta copies from LDS
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_25; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_variants.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
tools/build_variant.sh stamps -DCF_TRSM_STAMPS > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_stamps.so timeout -k 10 300 python tools/small_stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps.txt; head -6 $O/stamps.txt; tail -8 $O/stamps.txt
WS=1,16,32,48,64 REPS=400 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W=" | tee $O/wall.txt
cd /tmp && export TMPDIR=/tmp
WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace.log; exit 1; }
f=$(find $GRAFT_REPO_ROOT/$O/trace -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 600 | tee $GRAFT_REPO_ROOT/$O/kernels.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace
