#!/bin/bash
# round 4, GPU call 36: the small-blocks kernel as kept (sqrt_pos / div_pos, table-driven exp / log at the Gauss-Legendre nodes, nodes staged in LDS,
# column entries of the quadratic forms eight at a time; the BAO prefetch of call 35 dropped): whole suite, joint soaks, small-batch latencies
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_36; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "suite: $(tail -1 $O/pytest.log)"; grep -E "^E  |^FAILED" $O/pytest.log | cut -c1-300 | head -30
for wl in desi_cmb_des5y:cpl desi_cmb_des5y desi_cmb; do
  WORKLOAD=$wl CALLS=30000 timeout -k 10 400 python tools/soak_small_batches.py > $O/soak_$wl.txt 2>&1; tail -3 $O/soak_$wl.txt | cut -c1-400
done
WORKLOAD=desi_cmb_des5y:cpl WS=1,16,64,100,256,2048 REPS=300 timeout -k 10 300 python tools/small_batch_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/small_batch_wall_joint_cpl.txt
for cfg in "desi_cmb_des5y --fde cpl" "desi_cmb_des5y" "desi_des5y_bbn_theta_star"; do
  tag=$(echo $cfg | tr ' -' '__')
  python3 bench.py --workload $cfg > $O/bench_$tag.json 2>/dev/null && python -c "
import json; d=json.load(open('$O/bench_$tag.json')); c=d['cpu_baseline']; print('$tag', '%.4e'%d['value'], '%.4f'%d['ms_per_step'], d['kernels_ms'], 'parity %.1e'%c['parity_max_rel'])"
done
