#!/bin/bash
mkdir -p gpurun_out/prof6
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MI355XQR_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6 -o p -- python3 devtools/tools_perf.py 8192x8192x256 > gpurun_out/prof6/run.log 2>&1
python3 devtools/tools_trace_summary.py gpurun_out/prof6/p_kernel_trace.csv | sed 's/void //' | head -26
rm -f gpurun_out/prof6/p_kernel_trace.csv
