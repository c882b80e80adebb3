#!/bin/bash
# round 4, GPU call 20: the new bit-identity tests (knobs off vs on; half-unit batch sizes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_20; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_handoff.py tests/test_gpu_parity.py -m gpu -x -q -k "skipping or batch_invariance or release_ordered" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
