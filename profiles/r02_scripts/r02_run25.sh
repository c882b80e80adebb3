#!/bin/bash
# lean persistent walker kernel (late-time flat LCDM / thawing): parity, then A/B against the one-walker kernel (CF_WALKER_SLOTS=0)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/run25; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_fs8.py tests/test_scripts.py tests/test_variants.py tests/test_gpu_random_shapes.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
show() { python -c "
import json; d=json.load(open('$1')); print('$2', '%.4e'%d['value'], '%.4f ms'%d['ms_per_step'], d['kernels_ms'])"; }
for rep in 1 2 3; do
  CF_WALKER_SLOTS=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/one_$rep.json 2>/dev/null; show $O/one_$rep.json "one walker / wg "
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/pers_$rep.json 2>/dev/null; show $O/pers_$rep.json "persistent      "
done
for wl in "--walkers-per-gpu 8192" "--walkers-per-gpu 2048" "--walkers-per-gpu 1024"; do
  CF_WALKER_SLOTS=0 python3 bench.py $wl --no-cpu-baseline > $O/a.json 2>/dev/null; show $O/a.json "one walker / wg  $wl"
  python3 bench.py $wl --no-cpu-baseline > $O/b.json 2>/dev/null; show $O/b.json "persistent       $wl"
done
