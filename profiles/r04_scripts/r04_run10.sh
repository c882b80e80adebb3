#!/bin/bash
# round 4, GPU call 10: evidence pass on the static lean kernel: suite, driver's bench command, configs 3 / 5, the in-process k-replica
# mode (rehearsed on one GPU), in-kernel clock + bare MFMA loop on this box (JSON for bench.py to replay), PMC passes with durations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_10; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
timeout -k 10 120 tools/coexec_f64_rate > $O/bare_mfma_rate.txt 2>&1; head -4 $O/bare_mfma_rate.txt
OUT_JSON=$O/solve_clock.json BARE_TXT=$O/bare_mfma_rate.txt COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so timeout -k 10 300 python tools/solve_clock.py 1024 2048 4096 2>&1 | grep -v amdgpu.ids > $O/solve_clock.txt; grep "W =\|shader clock" $O/solve_clock.txt
mkdir -p profiles && cp $O/solve_clock.json profiles/r04_solve_clock.json
show() { python -c "
import json,sys
d=json.load(open('$1')); c=d.get('cpu_baseline',{}); r=d['roofline']; print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], {k:(round(v,4) if v else v) for k,v in d.get('kernels_ms',{}).items()}, 'frac %.3f'%r['frac'], r.get('frac_factors'), 'host-visible', d.get('value_host_visible'), 'cpu %.3e x%s parity %.1e'%(c.get('value',0),c.get('cores'),c.get('parity_max_rel',-1)))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }; show $O/bench.json
python3 bench.py --workload desi_cmb_des5y --no-cpu-baseline > $O/bench_config3_lcdm.json 2>/dev/null && show $O/bench_config3_lcdm.json
python3 bench.py --workload desi_cmb_des5y --fde cpl --no-cpu-baseline > $O/bench_config3_cpl.json 2>/dev/null && show $O/bench_config3_cpl.json
python3 bench.py --workload desi_des5y_bbn_theta_star --no-cpu-baseline > $O/bench_config5.json 2>/dev/null && show $O/bench_config5.json
python3 bench.py --mode inprocess --devices 0 --walkers-total 65536 --steps 10 --warmup 2 > $O/bench_inprocess_1.json 2> $O/bench_inprocess_1.err && show $O/bench_inprocess_1.json || tail -5 $O/bench_inprocess_1.err
python3 bench.py --mode inprocess --devices 0,0,0,0,0,0,0,0 --walkers-total 65536 --steps 10 --warmup 2 > $O/bench_inprocess_8on1.json 2> $O/bench_inprocess_8on1.err && show $O/bench_inprocess_8on1.json || tail -5 $O/bench_inprocess_8on1.err
CF_HOST_WAIT=block python3 bench.py --mode inprocess --devices 0,0,0,0,0,0,0,0 --walkers-total 65536 --steps 10 --warmup 2 > $O/bench_inprocess_8on1_block.json 2>/dev/null && show $O/bench_inprocess_8on1_block.json
python -c "
import json
for f in ('bench_inprocess_1','bench_inprocess_8on1','bench_inprocess_8on1_block'):
    d=json.load(open('$O/'+f+'.json')); print(f, 'ms/call %.3f'%d['ms_per_step'], 'host CPU-seconds per call %.4f'%d['host_cpu_seconds_per_call'])"
timeout -k 10 900 bash tools/pmc_profile.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-400
