#!/bin/bash
# same-box A/B of the last arriver's share loads: serial loop of dependent loads (variant build) vs side by side through LDS
cd $GRAFT_REPO_ROOT
O=gpurun_out/run19; mkdir -p $O
V=$GRAFT_REPO_ROOT/cosmology-model-fit_amd/libcosmofit_hip_serial.so
show() { python -c "
import json; d=json.load(open('$1')); print('$2', '%.4e'%d['value'], '%.4f ms'%d['ms_per_step'], 'solve %.4f ms'%d['kernels_ms']['tri_gemm_chi2_kernel'], 'frac %.3f'%d['roofline']['frac'])"; }
for rep in 1 2 3; do
  COSMOFIT_LIB=$V python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/serial_$rep.json 2>/dev/null; show $O/serial_$rep.json "serial  "
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/lds_$rep.json 2>/dev/null; show $O/lds_$rep.json "via LDS "
done
for w in 8192 16384; do
  COSMOFIT_LIB=$V python3 bench.py --walkers-per-gpu $w --no-cpu-baseline > $O/serial_w$w.json 2>/dev/null; show $O/serial_w$w.json "serial   W=$w"
  python3 bench.py --walkers-per-gpu $w --no-cpu-baseline > $O/lds_w$w.json 2>/dev/null; show $O/lds_w$w.json "via LDS  W=$w"
done
COSMOFIT_LIB=$V timeout -k 10 200 python tools/latency_probe.py 2>/dev/null | grep "inverse GEMM" | head -6 > $O/latency_serial.txt
timeout -k 10 200 python tools/latency_probe.py 2>/dev/null | grep "inverse GEMM" | head -6 > $O/latency_lds.txt
echo "== serial"; cat $O/latency_serial.txt; echo "== via LDS"; cat $O/latency_lds.txt
