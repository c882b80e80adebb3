#!/bin/bash
# round 4, GPU call 1: tests on the ADVICE fixes (incl. the no-SN soak), the round-3 kernel's solve time over batch sizes (the "before" of
# the work-queue kernel), and the alias A/B (timing builds, wrong results) at W = 8192 and 4096
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_1; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
WORKLOAD=desi_cmb CALLS=20000 timeout -k 10 300 python tools/soak_small_batches.py 2>&1 | grep -v amdgpu.ids | tee $O/soak_desi_cmb.txt
for rep in 1 2; do
  for W in 512 768 1024 1536 2048 3072 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh base_w${W}_$rep
  done
done 2>&1 | tee $O/base_sizes.txt
for rep in 1 2; do
  for W in 8192 4096; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh prod_w${W}_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_r04base.so
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh aliasA_w${W}_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_alias_a.so
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh aliasAB_w${W}_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_alias_ab.so
  done
done 2>&1 | tee $O/alias_ab.txt
