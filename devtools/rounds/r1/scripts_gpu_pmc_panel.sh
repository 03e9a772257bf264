#!/bin/bash
R=gpurun_out/pmc_panel; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mode in cholqr tsqr; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/${mode}_$ctr -o pmc -- python3 devtools/tools_pmc_panel.py $mode > $R/${mode}_${ctr}.json 2> $R/${mode}_$ctr.err
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/${mode}_trace -o tr -- python3 devtools/tools_pmc_panel.py $mode > $R/${mode}_trace.json 2> $R/${mode}_trace.err
f1=$(find $R/${mode}_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $R/${mode}_WRITE_SIZE -name "*counter_collection.csv" | head -1); f3=$(find $R/${mode}_trace -name "*kernel_trace.csv" | head -1)
python3 devtools/tools_pmc_panel_summary.py $f1 $f2 $f3 > $R/${mode}_hbm_summary.txt
rm -f $f1 $f2 $f3
cat $R/${mode}_hbm_summary.txt | head -40
done
