#!/bin/bash
# round 4, GPU call 25: walker kernel as two resident workgroups per CU looping over their walkers, against one workgroup per walker
# (same library, CF_TUNE walker_wgs_per_cu=0) and against the library before (wbase); parity suite first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_25; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
L=$PWD/cosmology-model-fit_amd
for rep in 1 2 3; do
  for cfg in "" "--workload desi_cmb_des5y --fde cpl" "--walkers-per-gpu 2048" "--walkers-per-gpu 1024"; do
    tag=$(echo "w4096 $cfg" | tr ' -' '__')
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_wbase_$rep COSMOFIT_LIB=$L/libcosmofit_hip_wbase.so
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_perwalker_$rep CF_TUNE=walker_wgs_per_cu=0
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_resident2_$rep
  done
done 2>&1 | tee $O/walker_resident_ab.txt
