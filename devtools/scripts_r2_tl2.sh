#!/bin/bash
# kernel timeline of the chain-bound phase with / without the incremental T:  scripts_r2_tl2.sh <shape>
S=${1:-8192x8192x256}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 0 1; do
R=gpurun_out/tli$c; rm -rf $R; mkdir -p $R
export MI355XQR_INCT=$c
rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o tl -- python3 devtools/tools_one.py $S > $R/log.txt 2>&1
f=$(find $R/t -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.50 2.5 > $R/timeline.txt
python3 devtools/tools_trace_summary.py $f > $R/summary.txt
rm -rf $R/t
done
