#!/bin/bash
# early-product leaf: kernel tests, full suite, timing table with / without (MI355XQR_EP=0)
set -o pipefail
mkdir -p gpurun_out/r3b
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "early_product" > gpurun_out/r3b/ep_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r3b/ep_tests.log; echo "ep tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/tests.log 2>&1; rc=$?; tail -5 gpurun_out/r3b/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
CFG="4096x512x128 1024x256x128 4096x4096x64 8192x8192x256 16384x16384x256 262144x512x128 65536x256x128 131072x256x128"
python3 devtools/tools_perf.py $CFG > gpurun_out/r3b/perf_ep1.txt 2>&1
MI355XQR_EP=0 python3 devtools/tools_perf.py $CFG > gpurun_out/r3b/perf_ep0.txt 2>&1
for f in perf_ep1 perf_ep0; do echo $f; cut -c1-110 gpurun_out/r3b/$f.txt; done
