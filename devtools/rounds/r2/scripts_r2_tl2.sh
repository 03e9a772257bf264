#!/bin/bash
# kernel timeline of one factorisation:  scripts_r2_tl2.sh <shape> <tag> [from-fraction] [window-ms]   (environment passes through)
S=${1:-8192x8192x256}; T=${2:-x}; F=${3:-0.50}; W=${4:-2.5}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=gpurun_out/tl_$T; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o tl -- python3 devtools/tools_one.py $S > $R/log.txt 2>&1
f=$(find $R/t -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f $F $W > $R/timeline.txt
python3 devtools/tools_trace_summary.py $f > $R/summary.txt
rm -rf $R/t
