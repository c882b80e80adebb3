#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
for shape in 2x2 1x2 2x2 1x2; do
  echo "== CF_GEMM_SHAPE=$shape"; CF_GEMM_SHAPE=$shape timeout -k 10 200 python tools/latency_probe.py 2>/dev/null | grep "inverse GEMM" | tee -a gpurun_out/r02p/latency_$shape.txt
done
