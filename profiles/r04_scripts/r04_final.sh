#!/bin/bash
# End-of-round evidence (round 4): GPU tests, smoke(), in-kernel clock + bare MFMA loop (the JSON bench.py replays), the driver's bench
# command, the other configs with their CPU legs, batch sizes, the in-process k-replica mode, small-batch wall times, the ensemble step,
# rocprofv3 kernel stats of the driver's command, PMC passes (counters WITH the dispatches' durations).  Outputs: gpurun_out/r04_final/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log | cut -c1-160
[ -x tools/coexec_f64_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/coexec_f64_rate.hip -o tools/coexec_f64_rate
[ -f cosmology-model-fit_amd/libcosmofit_hip_clock.so ] || tools/build_variant.sh clock -DCF_DIAG_CLOCK > /dev/null
timeout -k 10 120 tools/coexec_f64_rate > $O/bare_mfma_rate.txt 2>&1; head -4 $O/bare_mfma_rate.txt
OUT_JSON=$O/solve_clock.json BARE_TXT=$O/bare_mfma_rate.txt COSMOFIT_LIB=$PWD/cosmology-model-fit_amd/libcosmofit_hip_clock.so timeout -k 10 300 python tools/solve_clock.py 1024 2048 4096 2>&1 | grep -v amdgpu.ids > $O/solve_clock.txt; grep "W =\|shader clock\|idle CU" $O/solve_clock.txt
cp $O/solve_clock.json profiles/r04_solve_clock.json   # what bench.py replays as roofline.clock_ghz (also copied back by hand)
show() { python -c "
import json,sys
d=json.load(open('$1')); c=d.get('cpu_baseline',{}); r=d['roofline']; print('$1', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], {k:(round(v,4) if v else v) for k,v in d.get('kernels_ms',{}).items()}, 'frac %.3f'%r['frac'], {k:round(v,4) for k,v in (r.get('frac_factors') or {}).items()}, 'host-visible', d.get('value_host_visible'), 'cpu %.3e x%s parity %.1e'%(c.get('value',0),c.get('cores'),c.get('parity_max_rel',-1)))"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }; show $O/bench.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null && show $O/bench_default.json
for W in 512 1024 2048 8192; do python3 bench.py --walkers-per-gpu $W --no-cpu-baseline > $O/bench_w$W.json 2>/dev/null && show $O/bench_w$W.json; done
python3 bench.py --scaling strong --no-cpu-baseline --steps 10 > $O/bench_strong_w65536.json 2>/dev/null && show $O/bench_strong_w65536.json
python3 bench.py --workload desi_cmb_des5y > $O/bench_config3_lcdm.json 2>/dev/null && show $O/bench_config3_lcdm.json
python3 bench.py --workload desi_cmb_des5y --fde cpl > $O/bench_config3_cpl.json 2>/dev/null && show $O/bench_config3_cpl.json
python3 bench.py --workload desi_des5y_bbn_theta_star > $O/bench_config5.json 2>/dev/null && show $O/bench_config5.json
python3 bench.py --mode inprocess --devices 0 --steps 10 --warmup 2 > $O/bench_inprocess_1.json 2>/dev/null && show $O/bench_inprocess_1.json
python3 bench.py --mode inprocess --devices 0,0,0,0,0,0,0,0 --steps 10 --warmup 2 > $O/bench_inprocess_8on1.json 2>/dev/null && show $O/bench_inprocess_8on1.json
WS=1,16,32,64,75,128,150,256,512,1024,2048,4096 REPS=300 timeout -k 10 300 python tools/small_batch_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/small_batch_wall.txt
WORKLOAD=desi_cmb_des5y:cpl WS=1,16,64,100,256,2048 REPS=300 timeout -k 10 300 python tools/small_batch_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/small_batch_wall_joint_cpl.txt
timeout -k 10 300 python tools/ensemble_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/ensemble_step.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cut -c1-200 $O/kernel_stats.csv | head -8
rm -rf $O/prof
timeout -k 10 900 bash tools/pmc_profile.sh $O/pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-400
rm -rf $O/pmc/pass*/
