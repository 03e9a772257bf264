#!/bin/bash
# kernel timelines of the late, chain-bound phase of C3 (and of 8192^2) with the one-launch panel
R=gpurun_out/tl_r04; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o tl -- python3 devtools/tools_one.py 16384x16384x256 > $R/log.txt 2>&1
f=$(find $R/t -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.90 1.5 > $R/timeline_late.txt
python3 devtools/tools_trace_timeline.py $f 0.935 0.9 > $R/timeline_fused.txt
python3 devtools/tools_trace_timeline.py $f 0.985 0.8 > $R/timeline_last.txt
python3 devtools/tools_trace_summary.py $f > $R/summary.txt
python3 devtools/tools_gantt.py 16384x16384x256 > $R/gantt.txt 2>&1
rm -f $f
head -30 $R/summary.txt
