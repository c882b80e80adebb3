#!/bin/bash
# round 4, GPU call 37: 500 seeded random likelihood shapes against the numpy oracle (the round's kernel changes under shapes the suite's 24 seeds miss)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_37; mkdir -p $O
CF_TEST_RANDOM_SHAPES=500 timeout -k 10 1000 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q > $O/pytest.log 2>&1; echo "random shapes: $(tail -1 $O/pytest.log)"; grep -E "^E  |^FAILED" $O/pytest.log | cut -c1-300 | head -30
