#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_panel_cqr.py tests/test_gpu_factor32.py -x -q -m gpu > gpurun_out/cqr_test.txt 2>&1; echo "cqr tests rc=$?"; tail -3 gpurun_out/cqr_test.txt
python3 devtools/tools_cqr_debug.py 262144 128 stamps > gpurun_out/r5_cq_stamps.txt 2>&1
bash devtools/r4_cqr_e2.sh 262144 128 > gpurun_out/r5_cqr_kernel_times.txt 2>&1; cat gpurun_out/r5_cqr_kernel_times.txt
bash devtools/r5_sweep.sh > gpurun_out/r5_sweep.txt 2>&1
