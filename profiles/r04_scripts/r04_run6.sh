#!/bin/bash
# round 4, GPU call 6: can ONE wave per SIMD keep the matrix pipe busy with a deeper pipeline?  PF = 2 / 4 / 6 x 1-2 workgroups per CU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2_full or batch_invariance or golden" > $O/pytest_pf2.log 2>&1; tail -1 $O/pytest_pf2.log
CF_TUNE=gemm_pf=4 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2_full or batch_invariance or golden" > $O/pytest_pf4.log 2>&1; tail -1 $O/pytest_pf4.log
CF_TUNE=gemm_pf=6,gemm_np=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2_full or batch_invariance or golden" > $O/pytest_pf6.log 2>&1; tail -1 $O/pytest_pf6.log
for rep in 1 2; do
  for W in 512 1024 2048 4096; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_base_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_r04base.so
    for pf in 2 4 6; do for k in 1 2; do
      BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_pf${pf}k${k}_$rep CF_TUNE=gemm_wgs=$k,gemm_pf=$pf,gemm_np=2
    done; done
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_np1pf4k2_$rep CF_TUNE=gemm_wgs=2,gemm_pf=4,gemm_np=1
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_np1pf4k3_$rep CF_TUNE=gemm_wgs=3,gemm_pf=4,gemm_np=1
  done
done 2>&1 | tee $O/queue_sizes.txt
