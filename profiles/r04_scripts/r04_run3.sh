#!/bin/bash
# round 4, GPU call 3: software-pipelined work-queue solve kernel (v2): parity, then solve time over batch sizes x workgroups per
# CU against the round-3 kernel (libcosmofit_hip_r04base.so) on the same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_3; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2_full or batch_invariance or golden or configs3_shape" > $O/pytest_first.log 2>&1; rc=$?; tail -3 $O/pytest_first.log
[ $rc -ne 0 ] && { tail -40 $O/pytest_first.log; exit $rc; }
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
for rep in 1 2; do
  for W in 512 768 1024 1536 2048 3072 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_base_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_r04base.so
    for k in 0 1 2 3 4; do
      BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_k${k}_$rep CF_TUNE=gemm_wgs=$k
    done
  done
done 2>&1 | tee $O/queue_sizes.txt
