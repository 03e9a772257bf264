#!/bin/bash
R=gpurun_out/s2b; rm -rf $R; mkdir -p $R
python devtools/tools_gantt.py 16384x16384x256 > $R/gantt_c3.txt 2>&1
python devtools/tools_gantt.py 8192x8192x256 > $R/gantt_8k.txt 2>&1
head -70 $R/gantt_c3.txt
