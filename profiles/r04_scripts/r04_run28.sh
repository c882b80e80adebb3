#!/bin/bash
# round 4, GPU call 28: the whole GPU suite repeatedly (call 25 saw one wrong small batch in test_config2_random_small_batches_soak inside the full
# suite; 8 stand-alone repeats and 180,000 stand-alone soak calls saw none): production library (looping walker kernel) x3, library before (wbase) x2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_28; mkdir -p $O
L=$PWD/cosmology-model-fit_amd
for v in prod prod prod wbase wbase; do
  n=$((n+1))
  lib=$L/libcosmofit_hip_$v.so; [ $v = prod ] && lib=$L/libcosmofit_hip.so
  COSMOFIT_LIB=$lib timeout -k 10 600 python -m pytest tests -m gpu -q > $O/pytest_${n}_$v.log 2>&1
  echo "run $n $v: $(tail -1 $O/pytest_${n}_$v.log)"; grep -E "^E +Failed|^FAILED" $O/pytest_${n}_$v.log | cut -c1-600
done 2>&1 | tee $O/suite_repeats.txt
