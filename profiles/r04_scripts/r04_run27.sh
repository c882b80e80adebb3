#!/bin/bash
# round 4, GPU call 27: 60,000-call small-batch soaks of the looping walker kernel and its variants (call 25 saw ONE wrong batch in the 3000-call
# test; call 26's 8 repeats of that test saw none): which build mismatches, and whose walkers' results the wrong ones are
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_27; mkdir -p $O
L=$PWD/cosmology-model-fit_amd
for v in prod l0 wbase; do
  lib=$L/libcosmofit_hip_$v.so; [ $v = prod ] && lib=$L/libcosmofit_hip.so
  COSMOFIT_LIB=$lib CALLS=60000 timeout -k 10 300 python tools/soak_small_batches.py > $O/soak_$v.txt 2>&1
  echo "== $v"; tail -8 $O/soak_$v.txt | cut -c1-400
done 2>&1 | tee $O/soaks.txt
