#!/bin/bash
# round 4, GPU call 21: the in-process bench test; soak of the final kernels (random batch sizes 1 .. 3000, bitwise against the 4096 batch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_21; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rccl.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
CALLS=100000 timeout -k 10 600 python tools/soak_small_batches.py 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
WORKLOAD=desi_cmb_des5y:cpl CALLS=30000 timeout -k 10 400 python tools/soak_small_batches.py 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt
WORKLOAD=desi_cmb CALLS=50000 timeout -k 10 300 python tools/soak_small_batches.py 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt
