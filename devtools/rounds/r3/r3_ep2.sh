#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3d
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "early_product" > gpurun_out/r3d/ep_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r3d/ep_tests.log; echo "ep tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
CFG="4096x512x128 1024x256x128 4096x4096x64 8192x8192x256 16384x16384x256 262144x512x128 65536x256x128"
python3 devtools/tools_perf.py $CFG > gpurun_out/r3d/perf_ep1.txt 2>&1
MI355XQR_EP=0 python3 devtools/tools_perf.py $CFG > gpurun_out/r3d/perf_ep0.txt 2>&1
for f in perf_ep1 perf_ep0; do echo $f; cut -c1-100 gpurun_out/r3d/$f.txt; done
./devtools/rounds/r3/r3_prof.sh r3d_tall1 262144x512x128 | head -14
MI355XQR_EP=0 ./devtools/rounds/r3/r3_prof.sh r3d_tall0 262144x512x128 | head -14
./devtools/rounds/r3/r3_prof.sh r3d_sq1 8192x8192x256 | head -8
