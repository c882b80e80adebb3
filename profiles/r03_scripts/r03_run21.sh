#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_21; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_variants.py tests/test_gpu_joint.py tests/test_gpu_random_shapes.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
export BENCH_ARGS="--workload desi_cmb_des5y --fde cpl"
for rep in 1 2 3; do tools/quick_ab.sh c3cpl_$rep; done 2>&1 | tee $O/cpl.txt
