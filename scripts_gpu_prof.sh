#!/bin/bash
mkdir -p gpurun_out/prof3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MI355XQR_PANEL_CUS=32
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -o p -- python3 tools_perf.py 16384x16384x128 > gpurun_out/prof3/run.log 2>&1
python3 tools_trace_summary.py gpurun_out/prof3/p_kernel_trace.csv | sed 's/void //' | awk '{ if ($1 ~ /leaf_step/) { c+=$3; b+=$5 } else print } END { print "leaf_step_kernel<*> calls", c, "busy", b, "ms avg", b/c*1000, "us" }'
rm -f gpurun_out/prof3/p_kernel_trace.csv
