#!/bin/bash
# table build without the compiled-in fallbacks (pow / omnu_z) and with ln(1 + z) prefetched, against the previous build
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/run26; mkdir -p $O
V=$GRAFT_REPO_ROOT/cosmology-model-fit_amd/libcosmofit_hip_base.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joint.py tests/test_fs8.py tests/test_scripts.py tests/test_variants.py tests/test_gpu_random_shapes.py tests/test_plot_accessors.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
show() { python -c "
import json; d=json.load(open('$1')); print('$2', '%.4e'%d['value'], '%.4f ms'%d['ms_per_step'], d['kernels_ms'])"; }
for wl in "--workload desi_cmb_des5y" "--workload desi_cmb_des5y --fde cpl" "--workload desi_des5y_bbn_theta_star" "--gpus 1 --steps 20 --warmup 5"; do
  for rep in 1 2; do
    COSMOFIT_LIB=$V python3 bench.py $wl --no-cpu-baseline > $O/a.json 2>/dev/null; show $O/a.json "before  $wl"
    python3 bench.py $wl --no-cpu-baseline > $O/b.json 2>/dev/null; show $O/b.json "after   $wl"
  done
done
