#!/bin/bash
# NP = 4 (64 walkers per workgroup, 208 VGPRs, 2 waves per SIMD) against NP = 2 at large batches; bit-identity first
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/run21; mkdir -p $O
CF_GEMM_SHAPE=4x2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "invariance or config2 or config4 or ragged" > $O/pytest_np4.log 2>&1 || { tail -30 $O/pytest_np4.log; exit 1; }
tail -2 $O/pytest_np4.log
show() { python -c "
import json; d=json.load(open('$1')); print('$2', '%.4e'%d['value'], '%.4f ms'%d['ms_per_step'], 'solve %.4f ms'%d['kernels_ms']['tri_gemm_chi2_kernel'], 'frac %.3f'%d['roofline']['frac'])"; }
for w in 4096 8192 16384; do
  for rep in 1 2; do
  python3 bench.py --walkers-per-gpu $w --no-cpu-baseline > $O/np2_w${w}_$rep.json 2>/dev/null; show $O/np2_w${w}_$rep.json "NP=2 W=$w"
  CF_GEMM_SHAPE=4x2 python3 bench.py --walkers-per-gpu $w --no-cpu-baseline > $O/np4_w${w}_$rep.json 2>/dev/null; show $O/np4_w${w}_$rep.json "NP=4 W=$w"
  done
done
