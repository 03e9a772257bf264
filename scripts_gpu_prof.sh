#!/bin/bash
mkdir -p gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MI355XQR_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o p -- python3 tools_perf.py 4096x4096x128 > gpurun_out/prof2/run.log 2>&1
python3 tools_trace_summary.py gpurun_out/prof2/p_kernel_trace.csv
rm -f gpurun_out/prof2/p_kernel_trace.csv
