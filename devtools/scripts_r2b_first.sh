#!/bin/bash
# round-2 (second session) state check: gpu tests, perf table, tall-skinny kernel trace with every launch
mkdir -p gpurun_out
R=gpurun_out/s2a; rm -rf $R; mkdir -p $R
timeout -k 10 1100 python -m pytest tests -q -m gpu --timeout=600 -x > $R/tests.log 2>&1; echo "tests rc=$?"; tail -5 $R/tests.log
python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 262144x512x128 262144x256x128 2097152x512x128 > $R/perf.txt 2>&1
cat $R/perf.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/s -o tl -- python3 devtools/tools_one.py 262144x512x128 > $R/log_ts.txt 2>&1
f=$(find $R/s -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_timeline.py $f 0.50 8.0 > $R/timeline_ts.txt
python3 devtools/tools_trace_summary.py $f > $R/summary_ts.txt
rm -rf $R/s
head -40 $R/summary_ts.txt
