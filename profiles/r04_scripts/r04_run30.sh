#!/bin/bash
# round 4, GPU call 30: the committed walker kernel (SN loop with fewer instructions, one workgroup per walker): whole suite, long soaks of
# three workloads, A/B against the library before (wbase)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_30; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "suite: $(tail -1 $O/pytest.log)"; grep -E "^E +Failed|^FAILED" $O/pytest.log | cut -c1-700
for wl in pantheon desi_cmb_des5y desi_cmb_des5y:cpl desi_cmb; do
  WORKLOAD=$wl CALLS=100000 timeout -k 10 400 python tools/soak_small_batches.py > $O/soak_$wl.txt 2>&1; tail -4 $O/soak_$wl.txt | cut -c1-500
done
L=$PWD/cosmology-model-fit_amd
for rep in 1 2 3; do
  for cfg in "" "--workload desi_cmb_des5y --fde cpl"; do
    tag=$(echo "w4096 $cfg" | tr ' -' '__')
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_wbase_$rep COSMOFIT_LIB=$L/libcosmofit_hip_wbase.so
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_head_$rep
  done
done 2>&1 | tee $O/walker_head_ab.txt
