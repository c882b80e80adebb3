#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/tests; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
