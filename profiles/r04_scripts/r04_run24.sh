#!/bin/bash
# round 4, GPU call 24: walker kernel with fewer instructions -- SN loop (node i + 1 behind node i, polynomial constants in scalar
# registers, one range branch) and table build ((1 + z)^3 tabulated, three-address trapezoid fma) -- parity suite, then A/B against the
# library before (wbase), SN loop only (wsn), both (wall)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_24; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
L=$PWD/cosmology-model-fit_amd
for rep in 1 2 3; do
  for v in wbase wsn wall; do
    tools/quick_ab.sh w4096_${v}_$rep COSMOFIT_LIB=$L/libcosmofit_hip_$v.so
    BENCH_ARGS="--workload desi_cmb_des5y --fde cpl" tools/quick_ab.sh cpl_${v}_$rep COSMOFIT_LIB=$L/libcosmofit_hip_$v.so
  done
done 2>&1 | tee $O/walker_ab.txt
