#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 devtools/tools_cqr_debug.py 262144 128 stamps > gpurun_out/r5_cq_stamps.txt 2>&1
python3 devtools/tools_factor32.py > gpurun_out/f32_time.txt 2>&1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -5 gpurun_out/full_gpu.log
bash devtools/r5_base.sh > gpurun_out/r5_base2.txt 2>&1
