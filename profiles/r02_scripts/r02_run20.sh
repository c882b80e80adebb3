#!/bin/bash
# tile split TS = 2 (two workgroups per row block): bit-identity with it forced on, then solve time vs batch size off / on
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/run20; mkdir -p $O
CF_GEMM_TS_RANGE=1,1000000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_hard_cov.py -m gpu -x -q > $O/pytest_ts2.log 2>&1 || { tail -30 $O/pytest_ts2.log; exit 1; }
tail -2 $O/pytest_ts2.log
for rep in 1 2; do
CF_GEMM_TS_RANGE=1,0 timeout -k 10 200 python tools/latency_probe.py 2>/dev/null | grep "inverse GEMM" > $O/off_$rep.txt
CF_GEMM_TS_RANGE=1,1000000 timeout -k 10 200 python tools/latency_probe.py 2>/dev/null | grep "inverse GEMM" > $O/on_$rep.txt
echo "== off ($rep)"; cat $O/off_$rep.txt; echo "== TS = 2 ($rep)"; cat $O/on_$rep.txt
done
