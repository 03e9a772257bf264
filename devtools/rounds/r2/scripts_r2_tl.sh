#!/bin/bash
# kernel timelines (late, chain-bound phase and a tall-skinny shard) for one leaf generation: scripts_r2_tl.sh <leaf> 
L=${1:-2}
R=gpurun_out/tl$L; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MI355XQR_LEAF=$L
rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o tl -- python3 devtools/tools_one.py 16384x16384x256 > $R/log.txt 2>&1
f=$(find $R/t -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.90 1.5 > $R/timeline_late.txt
python3 devtools/tools_trace_summary.py $f > $R/summary.txt
rm -f $f
rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o tl -- python3 devtools/tools_one.py 262144x512x128 > $R/log_ts.txt 2>&1
f=$(find $R/s -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.60 1.2 > $R/timeline_ts.txt
python3 devtools/tools_trace_summary.py $f > $R/summary_ts.txt
rm -f $f
head -30 $R/summary.txt
