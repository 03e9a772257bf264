#!/bin/bash
# kernel summary of one 8192^2 factorisation (chain-bound): scripts_r2b_tl8k.sh <tag> [ENV=...]
T=$1; shift
R=gpurun_out/tl8k_$T; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $R/t -o tl -- python3 devtools/tools_one.py 8192x8192x256 > $R/log.txt 2>&1
f=$(find $R/t -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.75 0.35 > $R/timeline.txt
python3 devtools/tools_trace_summary.py $f > $R/summary.txt
rm -rf $R/t
head -24 $R/summary.txt
