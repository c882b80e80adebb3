#!/bin/bash
# round 4, GPU call 26: which part of the looping walker kernel breaks the small-batch soak (call 25: W = 6 wrong after 241 calls)?
# variants: l0 no opaque asm, l1 arguments re-read only, l2 thread index re-taken only, production (both)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_26; mkdir -p $O
L=$PWD/cosmology-model-fit_amd
for v in l0 l1 l2 prod; do
  lib=$L/libcosmofit_hip_$v.so; [ $v = prod ] && lib=$L/libcosmofit_hip.so
  for rep in 1 2; do
    COSMOFIT_LIB=$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "soak or batch_invariance" > $O/pytest_${v}_$rep.log 2>&1
    echo "$v rep $rep: $(tail -1 $O/pytest_${v}_$rep.log)"; grep -E "^E +W=|Mismatched" $O/pytest_${v}_$rep.log | head -3
  done
done 2>&1 | tee $O/bisect.txt
