#!/bin/bash
# round 6, run 33: C2 (4096^2 at nb 64) with the look-ahead schedule forced: where does a step's time go?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run33; mkdir -p $O
MI355XQR_LOOKAHEAD=1 python3 devtools/tools_gantt.py 4096x4096x64 2>/dev/null > $O/gantt_c2_la.txt
MI355XQR_LOOKAHEAD=1 MI355XQR_SPLIT=32 python3 devtools/tools_gantt.py 4096x4096x64 2>/dev/null > $O/gantt_c2_la32.txt
head -30 $O/gantt_c2_la.txt; sed -n 40,50p $O/gantt_c2_la.txt; head -14 $O/gantt_c2_la32.txt
