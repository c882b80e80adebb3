#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
for mode in spin block spin block; do
  echo "== CF_HOST_WAIT=$mode"; CF_HOST_WAIT=$mode timeout -k 10 200 python tools/latency_probe.py 2>/dev/null | grep "inverse GEMM" | tee -a $O/latency_$mode.txt
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['value_host_visible'])"
python -c "import __graft_entry__ as g; g.smoke()"
