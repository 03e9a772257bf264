#!/bin/bash
# round 6, run 2: kernel-level view of the tail of C3 (what sits between two one-launch panels) and of a refused full-width panel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/c3 -o tl -- python3 devtools/tools_perf.py 16384x16384x256 > $O/c3.log 2>&1
T=$(find $O/c3 -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_window.py $T 12 10 > $O/c3_tail_steps_52_53.txt
python3 devtools/tools_trace_window.py $T 30 28 > $O/c3_tail_steps_34_35.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cond -o tl -- python3 devtools/tools_cond.py 1e9 262144x512x128 > $O/cond.log 2>&1
T2=$(find $O/cond -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_trace_window.py $T2 2 1 cqr_gram_kernel > $O/cond_last_panel.txt
cp $(find $O/cond -name "*kernel_stats.csv" | head -1) $O/cond_kernel_stats.csv
python3 devtools/tools_cond.py 0 262144x512x128 65536x256x128 > $O/price.txt 2>&1
python3 devtools/tools_cond.py 1e9 262144x512x128 65536x256x128 >> $O/price.txt 2>&1
python3 devtools/tools_cond.py 1e5 262144x512x128 65536x256x128 >> $O/price.txt 2>&1
cat $O/price.txt
rm -rf $O/c3 $O/cond
