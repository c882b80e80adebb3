#!/bin/bash
# round 4, GPU call 15: the distributed bench tests (RCCL one rank, gloo two ranks on one GPU), bench lines refreshed (traffic replayed from the corrected PMC file)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_15; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rccl.py -m gpu -x -q > $O/pytest_rccl.log 2>&1; tail -15 $O/pytest_rccl.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('%.4e evals/s %.4f ms frac %.3f traffic %.1f MB %s' % (d['value'], d['ms_per_step'], r['frac'], r['traffic']/1e6, r['frac_factors']))"
BENCH_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_gloo2_rehearsal.json 2> $O/bench_gloo2.err || tail -5 $O/bench_gloo2.err
python -c "
import json; d=json.load(open('$O/bench_gloo2_rehearsal.json')); print(d['n_gpus'], d['ms_per_step'], d['ms_per_step_by_rank'], d['config']['parallelism'])"
