#!/bin/bash
# round 6, run 31: Gantt of 64-CU-split look-ahead shapes with one-launch tall panels: is the panel stream idle (balance model's chain time too long)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run31; mkdir -p $O
python3 devtools/tools_gantt.py 16384x8192x256 2>/dev/null > $O/gantt_16384x8192.txt
python3 devtools/tools_gantt.py 12288x4096x256 2>/dev/null > $O/gantt_12288x4096.txt
MI355XQR_SPLIT=64 python3 devtools/tools_gantt.py 16384x16384x256 2>/dev/null > $O/gantt_c3_split64.txt
head -36 $O/gantt_16384x8192.txt; head -20 $O/gantt_12288x4096.txt; head -36 $O/gantt_c3_split64.txt
