#!/bin/bash
# round 4, GPU call 40: the committed tree as the driver will see it: build(), GPU suite, smoke(), the driver's bench command, the example
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_40; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-160
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-120
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('%.4e evals/s, %.4f ms/step, frac %.3f'%(d['value'], d['ms_per_step'], d['roofline']['frac']))"
for f in examples/*.py; do timeout -k 10 300 python $f > $O/$(basename $f).log 2>&1; echo "$f rc=$?: $(tail -1 $O/$(basename $f).log | cut -c1-200)"; done
