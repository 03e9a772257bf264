#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_qr.py -q -m gpu --timeout=300 -x > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/tests.log
timeout 900 python tools_perf.py 4096x4096x128 8192x8192x128 16384x16384x128 16384x16384x256 262144x512x128 65536x256x128 > gpurun_out/perf2.log 2>&1
MI355XQR_LOOKAHEAD=0 timeout 900 python tools_perf.py 4096x4096x128 16384x16384x128 >> gpurun_out/perf2.log 2>&1
grep -v amdgpu.ids gpurun_out/perf2.log
mkdir -p gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MI355XQR_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o p -- python3 tools_perf.py 4096x4096x128 > gpurun_out/prof2/run.log 2>&1
python3 tools_trace_summary.py gpurun_out/prof2/p_kernel_trace.csv
rm -f gpurun_out/prof2/p_kernel_trace.csv
