#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -30 $O/pytest.log
