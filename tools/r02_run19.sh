#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
timeout -k 10 600 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q > gpurun_out/r02o/pytest_random.log 2>&1; tail -40 gpurun_out/r02o/pytest_random.log
