#!/bin/bash
R=gpurun_out/tl; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/t -o tl -- python3 devtools/tools_one.py 16384x16384x256 > $R/log.txt 2>&1
f=$(find $R/t -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.90 2.0 > $R/timeline_late.txt
python3 devtools/tools_trace_timeline.py $f 0.56 12 > $R/timeline_early.txt
rm -f $f
wc -l $R/timeline_late.txt $R/timeline_early.txt
