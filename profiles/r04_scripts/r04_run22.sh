#!/bin/bash
# round 4, GPU call 22: what the driver runs at round end: the whole GPU suite, smoke(), the bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_22; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -60 $O/pytest.log; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('%.4e evals/s %.4f ms frac %.3f' % (d['value'], d['ms_per_step'], r['frac']), {k:round(v,3) for k,v in r['frac_factors'].items()})"
