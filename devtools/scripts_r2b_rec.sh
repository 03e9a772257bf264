#!/bin/bash
R=gpurun_out/s2d; rm -rf $R; mkdir -p $R
python devtools/tools_records.py 16384x16384x256 30 46 > $R/rec_early.txt 2>&1
MI355XQR_EARLY_NEXT=0 python devtools/tools_records.py 16384x16384x256 30 46 > $R/rec_noearly.txt 2>&1
cat $R/rec_early.txt
