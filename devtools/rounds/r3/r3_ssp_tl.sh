#!/bin/bash
R=gpurun_out/ssp_tl; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o tl -- python3 devtools/tools_one.py 262144x512x128 > $R/log_ts.txt 2>&1
f=$(find $R/s -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.60 1.2 > $R/timeline_ts.txt
python3 devtools/tools_trace_summary.py $f > $R/summary_ts.txt
rm -f $f
