#!/bin/bash
# round 3, GPU call 53: where should the throughput solve kernel switch from 16- to 32-walker panels? (variant builds: from 449 / 513 walkers; shipped: 577)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_53; mkdir -p $O
tools/build_variant.sh np448 -DCF_NP2_FROM=448 > $O/b1.log 2>&1 || { tail $O/b1.log; exit 1; }
tools/build_variant.sh np512 -DCF_NP2_FROM=512 > $O/b2.log 2>&1 || { tail $O/b2.log; exit 1; }
for rep in 1 2; do for v in "" np512 np448; do
  echo "== variant ${v:-shipped (577)}"
  if [ -z "$v" ]; then WS=448,480,512,528,544,560,576 REPS=200 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="; else COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_$v.so WS=448,480,512,528,544,560,576 REPS=200 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="; fi
done; done | tee $O/wall.txt
