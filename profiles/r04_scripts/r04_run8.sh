#!/bin/bash
# round 4, GPU call 8: s_setprio 3 outside the K loop (prologue, exchange, hand-off) against the same kernel without it, same box; timeline with it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_8; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2_full or batch_invariance or golden" > $O/pytest.log 2>&1; tail -1 $O/pytest.log
for rep in 1 2 3; do
  for W in 512 1024 2048 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_noprio_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_noprio.so
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_prio_$rep
  done
done 2>&1 | tee $O/prio_ab.txt
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so timeout -k 10 300 python tools/solve_clock.py 1024 2048 4096 2>&1 | grep -v amdgpu.ids | tee $O/solve_clock.txt
