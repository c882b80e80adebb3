#!/bin/bash
# round 4, GPU call 11: the analytic pins of the three moves on the kernels' outputs; host wait policy (CPU-seconds per call of an 8-replica handle)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_11; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_moves_analytic.py -m gpu -x -q > $O/pytest_moves.log 2>&1; tail -15 $O/pytest_moves.log
for mode in default spin block; do
  for devs in 0 0,0,0,0,0,0,0,0; do
    if [ $mode = default ]; then python3 bench.py --mode inprocess --devices $devs --walkers-total 65536 --steps 20 --warmup 3 > $O/inproc_${mode}_${devs//,/}.json 2>/dev/null
    else CF_HOST_WAIT=$mode python3 bench.py --mode inprocess --devices $devs --walkers-total 65536 --steps 20 --warmup 3 > $O/inproc_${mode}_${devs//,/}.json 2>/dev/null; fi
    python -c "
import json; d=json.load(open('$O/inproc_${mode}_${devs//,/}.json')); print('CF_HOST_WAIT=$mode replicas', len(d['config']['replicas']), ': %.3f ms per 65536-walker call, %.2e evals/s, host CPU %.2f ms per call' % (d['ms_per_step'], d['value'], 1e3*d['host_cpu_seconds_per_call']))"
  done
done 2>&1 | tee $O/host_wait.txt
WS=16,75,512,4096 REPS=300 timeout -k 10 300 python tools/small_batch_timeline.py 2>&1 | grep "W=" | tee -a $O/host_wait.txt
