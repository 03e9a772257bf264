#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3g
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "guard or cholqr or early_product" > gpurun_out/r3g/tests.log 2>&1; rc=$?; tail -6 gpurun_out/r3g/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
CFG="262144x512x128 65536x256x128 131072x256x128 2097152x512x128"
python3 devtools/tools_perf.py $CFG > gpurun_out/r3g/perf_c1.txt 2>&1
MI355XQR_TALL_COOP=0 python3 devtools/tools_perf.py $CFG > gpurun_out/r3g/perf_c0.txt 2>&1
for f in perf_c1 perf_c0; do echo $f; cut -c1-100 gpurun_out/r3g/$f.txt | grep -v amdgpu; done
