#!/bin/bash
# round-2 final evidence run: GPU suite (with the printed parity numbers), the driver's bench command, rocprof stats, all configurations, probes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "hard_cov or fs8" > $O/pytest_hard_fs8.log 2>&1; grep -E "solve=|passed|failed" $O/pytest_hard_fs8.log | tail -12
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'], d['roofline']['traffic_source'], d.get('value_host_visible'))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; show $O/bench.json
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err; cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof; head -4 $O/kernel_stats.csv
python3 bench.py --walkers-per-gpu 8192 --no-cpu-baseline > $O/bench_w8192.json 2>/dev/null; show $O/bench_w8192.json
python3 bench.py --scaling strong --no-cpu-baseline --steps 10 > $O/bench_strong_w65536.json 2>/dev/null; show $O/bench_strong_w65536.json
python3 bench.py --workload desi_cmb_des5y > $O/bench_config3_lcdm.json 2>/dev/null; show $O/bench_config3_lcdm.json
python3 bench.py --workload desi_cmb_des5y --fde cpl > $O/bench_config3_cpl.json 2>/dev/null; show $O/bench_config3_cpl.json
python3 bench.py --workload desi_des5y_bbn_theta_star > $O/bench_config5.json 2>/dev/null; show $O/bench_config5.json
python3 bench.py --solve blocked --no-cpu-baseline > $O/bench_blocked.json 2>/dev/null; show $O/bench_blocked.json
timeout -k 10 200 python tools/fs8_probe.py > $O/fs8_probe.txt 2>&1; cat $O/fs8_probe.txt
timeout -k 10 300 python tools/ensemble_probe.py > $O/ensemble_probe.txt 2>&1; cat $O/ensemble_probe.txt
timeout -k 10 300 python tools/latency_probe.py > $O/latency_probe.txt 2>&1; tail -8 $O/latency_probe.txt
