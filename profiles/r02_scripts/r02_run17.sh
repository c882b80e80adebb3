#!/bin/bash
# growth kernel as a scan of 2 x 2 step matrices: parity tests, then the per-call time and the kernel's own time
set -e
mkdir -p gpurun_out/run17
timeout -k 10 600 python -m pytest tests/test_fs8.py tests/test_scripts.py -m gpu -x -q -s > gpurun_out/run17/pytest.log 2>&1 || { tail -40 gpurun_out/run17/pytest.log; exit 1; }
tail -5 gpurun_out/run17/pytest.log
timeout -k 10 300 python tools/fs8_probe.py > gpurun_out/run17/fs8_probe.txt 2>&1
cat gpurun_out/run17/fs8_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/run17/prof -- python3 $GRAFT_REPO_ROOT/tools/fs8_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/run17/prof -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-160 | head -12
