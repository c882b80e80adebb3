#!/bin/bash
# round 4, GPU call 13: dark-energy exp chained across a thread's nodes (table build, wCDM / CPL) against an exp per node: parity, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_13; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -60 $O/pytest.log; exit $rc; }
for rep in 1 2 3; do
  BENCH_ARGS="--workload desi_cmb_des5y --fde cpl" tools/quick_ab.sh c3cpl_nochain_$rep COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_nochain.so
  BENCH_ARGS="--workload desi_cmb_des5y --fde cpl" tools/quick_ab.sh c3cpl_chain_$rep
done 2>&1 | tee $O/fde_chain_ab.txt
