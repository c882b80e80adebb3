#!/bin/bash
# last arriver of the solve: shares fetched side by side through LDS -- parity / invariance tests, latency vs batch size, bench
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/run18; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_hard_cov.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/latency_probe.py 2>/dev/null | grep "inverse GEMM" > $O/latency.txt; cat $O/latency.txt
timeout -k 10 200 python tools/zerocopy_probe.py 2>/dev/null > $O/hostcalls.txt; cat $O/hostcalls.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench.json')); print('%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'], d.get('value_host_visible'))"
