#!/bin/bash
# End-of-round evidence (round 3): GPU tests, smoke(), the driver's bench command, the other configs with their CPU legs, the growth
# parity probe, the small-batch timeline, rocprofv3 kernel stats of the driver's command.  Outputs under gpurun_out/r03_final/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log | cut -c1-160
show() { python -c "
import json,sys
d=json.load(open('$1')); c=d.get('cpu_baseline',{}); print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], {k:(round(v,4) if v else v) for k,v in d['kernels_ms'].items()}, 'frac %.3f'%d['roofline']['frac'], 'host-visible', d.get('value_host_visible'), 'cpu %.3e x%s parity %.1e'%(c.get('value',0),c.get('cores'),c.get('parity_max_rel',-1)))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }; show $O/bench.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null && show $O/bench_default.json
python3 bench.py --walkers-per-gpu 8192 --no-cpu-baseline > $O/bench_w8192.json 2>/dev/null && show $O/bench_w8192.json
python3 bench.py --workload desi_cmb_des5y > $O/bench_config3_lcdm.json 2>/dev/null && show $O/bench_config3_lcdm.json
python3 bench.py --workload desi_cmb_des5y --fde cpl > $O/bench_config3_cpl.json 2>/dev/null && show $O/bench_config3_cpl.json
python3 bench.py --workload desi_des5y_bbn_theta_star > $O/bench_config5.json 2>/dev/null && show $O/bench_config5.json
timeout -k 10 300 python tools/fs8_parity_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/fs8_parity.txt
WS=1,16,32,64,75,128,150,256,512,2048 REPS=300 timeout -k 10 300 python tools/small_batch_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/small_batch_wall.txt
WORKLOAD=desi_cmb_des5y:cpl WS=1,16,64,100,256,2048 REPS=300 timeout -k 10 300 python tools/small_batch_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/small_batch_wall_joint_cpl.txt
cd /tmp && export TMPDIR=/tmp
WS=16 REPS=300 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/small_batch_timeline.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/trace -name '*kernel_trace.csv' | head -1); python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 600 | tee $GRAFT_REPO_ROOT/$O/timeline_w16.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cut -c1-200 $O/kernel_stats.csv | head -8
rm -rf $O/prof $O/trace
