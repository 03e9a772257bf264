#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3f
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "tn_update_wide or splitk or gemm_tn" > gpurun_out/r3f/tn_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r3f/tn_tests.log; echo "tn tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -m pytest tests/test_gpu_multipanel_golden.py -x -q > gpurun_out/r3f/golden.log 2>&1; rc=$?; tail -3 gpurun_out/r3f/golden.log; echo "golden rc=$rc"
[ $rc -ne 0 ] && exit $rc
CFG="8192x8192x256 16384x16384x256 16384x16384x512 12288x12288x256"
python3 devtools/tools_perf.py $CFG > gpurun_out/r3f/perf_w1.txt 2>&1
MI355XQR_TN_WIDE=0 python3 devtools/tools_perf.py $CFG > gpurun_out/r3f/perf_w0.txt 2>&1
for f in perf_w1 perf_w0; do echo $f; cut -c1-330 gpurun_out/r3f/$f.txt | grep -v amdgpu; done
