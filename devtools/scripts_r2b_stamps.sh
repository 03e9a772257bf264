#!/bin/bash
R=gpurun_out/s2g; rm -rf $R; mkdir -p $R
python devtools/tools_leaf_stamps.py 6144 > $R/stamps_6144.txt 2>&1
python devtools/tools_leaf_stamps.py 2048 > $R/stamps_2048.txt 2>&1
cat $R/stamps_6144.txt $R/stamps_2048.txt
