#!/bin/bash
R=gpurun_out/s2o; rm -rf $R; mkdir -p $R
MI355XQR_SPLIT=32:0.45,64 python devtools/tools_gantt.py 16384x16384x256 > $R/gantt_split.txt 2>&1
python devtools/tools_gantt.py 16384x16384x256 > $R/gantt_def.txt 2>&1
sed -n 30,70p $R/gantt_split.txt | cut -c1-140
