#!/bin/bash
# round-2 PMC passes: bench configurations whose roofline.traffic bench.py replays from profiles/r02_pmc_traffic_*.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02pmc; mkdir -p $O
run() { name=$1; shift; echo "== $name: $*"; bash tools/pmc_profile.sh $O/$name "$@" > $O/$name.log 2>&1; tail -3 $O/$name.log; cp $O/$name/pmc_traffic.json $O/pmc_traffic_$name.json; cp $O/$name/pmc_summary.txt $O/pmc_summary_$name.txt; rm -rf $O/$name/pass*/; }
run config2_w4096
run config4_w8192 --walkers-per-gpu 8192
run config3_lcdm --workload desi_cmb_des5y
run config3_cpl --workload desi_cmb_des5y --fde cpl
run config5 --workload desi_des5y_bbn_theta_star
# plain stats of the headline configuration and the benches of every configuration (unprofiled)
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 50 > $GRAFT_REPO_ROOT/$O/prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err; cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
timeout -k 10 300 python bench.py > $O/bench_config2.json 2> $O/bench_config2.err
timeout -k 10 300 python bench.py --walkers-per-gpu 8192 --no-cpu-baseline > $O/bench_w8192.json 2>/dev/null
timeout -k 10 300 python bench.py --scaling strong --no-cpu-baseline --steps 10 > $O/bench_strong_w65536.json 2>/dev/null
timeout -k 10 300 python bench.py --workload desi_cmb_des5y > $O/bench_config3_lcdm.json 2>/dev/null
timeout -k 10 300 python bench.py --workload desi_cmb_des5y --fde cpl > $O/bench_config3_cpl.json 2>/dev/null
timeout -k 10 300 python bench.py --workload desi_des5y_bbn_theta_star > $O/bench_config5.json 2>/dev/null
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'], d['roofline']['traffic_source'])"; done
timeout -k 10 200 python tools/event_overhead_probe.py > $O/event_overhead_probe.txt 2>&1; cat $O/event_overhead_probe.txt
