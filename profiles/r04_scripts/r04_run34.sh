#!/bin/bash
# round 4, GPU call 34: small-blocks kernel, + Gauss-Legendre nodes staged in LDS and the quadratic forms' column entries loaded eight at a
# time (the kernel waited half its life in s_waitcnt, call 33): joint tests, A/B against the library before all of it (sbbase), PMC after
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_34; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_joint.py tests/test_gpu_script.py tests/test_gpu_random_shapes.py tests/test_gpu_golden.py -m gpu -q > $O/pytest.log 2>&1; echo "joint tests: $(tail -1 $O/pytest.log)"; grep -E "^E  |^FAILED" $O/pytest.log | cut -c1-300 | head -30
L=$PWD/cosmology-model-fit_amd/libcosmofit_hip_sbbase.so
for rep in 1 2 3; do
  for cfg in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y" "desi_des5y_bbn_theta_star"; do
    tag=$(echo $cfg | tr ' -' '__')
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh gl_${tag}_before_$rep COSMOFIT_LIB=$L
    BENCH_ARGS="--workload $cfg" tools/quick_ab.sh gl_${tag}_after_$rep
  done
done 2>&1 | tee $O/gl_nodes_ab.txt
timeout -k 10 900 bash tools/pmc_profile.sh $O/pmc --workload desi_cmb_des5y --fde cpl > $O/pmc.log 2>&1
rm -rf $O/pmc/pass*/
grep -A32 "small_blocks" $O/pmc/pmc_summary.txt | grep "GUI_ACTIVE\|INSTS_VALU \|INSTS_VMEM_RD\|WAIT_ANY\|WAVE_CYCLES\|ACTIVE_INST_VALU"
