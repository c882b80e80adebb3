#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
timeout -k 10 300 python tools/occupancy_probe.py 2>/dev/null | tee gpurun_out/r02m/occupancy_probe.txt
