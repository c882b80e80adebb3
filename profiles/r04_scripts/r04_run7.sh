#!/bin/bash
# round 4, GPU call 7: the static-grid solve kernel on lean arguments (0 SGPR spills, 122 VGPRs): whole suite, same-box A/B against the
# round-3 build over batch sizes, its in-kernel timeline (diagnostic build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_7; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
for rep in 1 2 3; do
  for W in 512 1024 2048 4096 8192; do
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_r3_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_r04base.so
    BENCH_ARGS="--walkers-per-gpu $W" tools/quick_ab.sh q_w${W}_lean_$rep
  done
done 2>&1 | tee $O/lean_ab.txt
COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so timeout -k 10 300 python tools/solve_clock.py 1024 2048 4096 8192 2>&1 | grep -v amdgpu.ids | tee $O/solve_clock.txt
