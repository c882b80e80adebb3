#!/bin/bash
# round 4, GPU call 32: small-blocks kernel with H(z) at the Gauss-Legendre nodes through sqrt_pos / div_pos (the library's bits without its
# scaling selects) and a table-driven exp / log for the dark-energy factor: whole suite, then A/B against the library before (sbbase)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_32; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "suite: $(tail -1 $O/pytest.log)"; grep -E "^E  |^FAILED" $O/pytest.log | cut -c1-300 | head -30
L=$PWD/cosmology-model-fit_amd/libcosmofit_hip_sbbase.so
for rep in 1 2 3; do
  for cfg in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y" "desi_des5y_bbn_theta_star"; do
    tag=$(echo $cfg | tr ' -' '__')
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh gl_${tag}_before_$rep COSMOFIT_LIB=$L
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh gl_${tag}_after_$rep
  done
done 2>&1 | tee $O/gl_nodes_ab.txt
