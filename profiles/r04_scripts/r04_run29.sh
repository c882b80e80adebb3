#!/bin/bash
# round 4, GPU call 29: walker kernel as two resident workgroups per CU looping over their walkers (production) against one workgroup
# per walker (same library, CF_TUNE walker_wgs_per_cu=0) and against the library before the loop (wbase)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_29; mkdir -p $O
L=$PWD/cosmology-model-fit_amd
for rep in 1 2 3; do
  for cfg in "" "--workload desi_cmb_des5y --fde cpl" "--walkers-per-gpu 2048" "--walkers-per-gpu 1024"; do
    tag=$(echo "w4096 $cfg" | tr ' -' '__')
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_wbase_$rep COSMOFIT_LIB=$L/libcosmofit_hip_wbase.so
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_perwalker_$rep CF_TUNE=walker_wgs_per_cu=0
    BENCH_ARGS="$cfg" tools/quick_ab.sh ${tag}_resident2_$rep
  done
done 2>&1 | tee $O/walker_resident_ab.txt
