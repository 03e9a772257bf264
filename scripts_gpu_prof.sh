#!/bin/bash
mkdir -p gpurun_out/prof5
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MI355XQR_PANEL=tsqr rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -o p -- python3 tools_perf.py 262144x512x128 > gpurun_out/prof5/run.log 2>&1
python3 tools_trace_summary.py gpurun_out/prof5/p_kernel_trace.csv | sed 's/void //' | head -24
rm -f gpurun_out/prof5/p_kernel_trace.csv
