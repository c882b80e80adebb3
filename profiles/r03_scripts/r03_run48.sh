#!/bin/bash
# round 3, GPU call 48: snake order of the throughput solve kernel's workgroups for batches that are resident all at once (W = 256..2048)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_48; mkdir -p $O
tools/build_variant.sh snake -DCF_GEMM_SNAKE > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
for rep in 1 2; do
  echo "== descending (shipped)"; WS=256,384,512,768,1024,2048,4096 REPS=200 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
  echo "== snake"; COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_snake.so WS=256,384,512,768,1024,2048,4096 REPS=200 timeout -k 10 200 python tools/small_batch_timeline.py 2>&1 | grep "W="
done | tee $O/wall.txt
