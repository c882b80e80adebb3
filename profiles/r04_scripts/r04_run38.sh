#!/bin/bash
# round 4, GPU call 38: tests/test_gpu_parity.py (the file whose 3000-call small-batch test once saw a wrong batch with the dropped looping
# build) twelve times over with the final library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_38; mkdir -p $O
for n in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q > $O/pytest_$n.log 2>&1
  echo "repeat $n: $(tail -1 $O/pytest_$n.log)"; grep -E "^E +Failed|^FAILED" $O/pytest_$n.log | cut -c1-700
done 2>&1 | tee $O/parity_repeats.txt
