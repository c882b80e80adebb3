#!/bin/bash
# End-of-round evidence: smoke(), the driver's bench command, the flag-less default, the other configs, and the rocprofv3
# kernel stats of the driver's command.  Outputs under gpurun_out/final/ (copied into profiles/ by hand).
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'], d.get('value_host_visible'))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }; show $O/bench.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null && show $O/bench_default.json
python3 bench.py --walkers-per-gpu 8192 --no-cpu-baseline > $O/bench_w8192.json 2>/dev/null && show $O/bench_w8192.json
python3 bench.py --scaling strong --no-cpu-baseline --steps 10 > $O/bench_strong_w65536.json 2>/dev/null && show $O/bench_strong_w65536.json
python3 bench.py --solve blocked --no-cpu-baseline > $O/bench_blocked_solve.json 2>/dev/null && show $O/bench_blocked_solve.json
python3 bench.py --workload desi_cmb_des5y --no-cpu-baseline > $O/bench_config3_lcdm.json 2>/dev/null && show $O/bench_config3_lcdm.json
python3 bench.py --workload desi_cmb_des5y --fde cpl --no-cpu-baseline > $O/bench_config3_cpl.json 2>/dev/null && show $O/bench_config3_cpl.json
python3 bench.py --workload desi_des5y_bbn_theta_star --no-cpu-baseline > $O/bench_config5.json 2>/dev/null && show $O/bench_config5.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cut -c1-200 $O/kernel_stats.csv | head -8
