#!/bin/bash
cd $GRAFT_REPO_ROOT
SWEEP_ONLY=1 python3 devtools/r5_guard_sweep.py 2>&1 | grep -v amdgpu | cut -c1-60
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -5 gpurun_out/full_gpu.log
bash devtools/r5_base.sh > gpurun_out/r5_base2.txt 2>&1
