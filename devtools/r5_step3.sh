#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -5 gpurun_out/full_gpu.log
bash devtools/r5_base.sh > gpurun_out/r5_base2.txt 2>&1
