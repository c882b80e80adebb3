#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['from_idle'], d['kernels_ms'], d['roofline']['frac'], d['kernel_timing'], d['value_host_visible'])"
python3 bench.py > $O/bench_default.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['from_idle']['value'], d['kernels_ms'], d['roofline']['frac'])"
