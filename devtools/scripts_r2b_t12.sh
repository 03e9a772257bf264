#!/bin/bash
R=gpurun_out/s2m; rm -rf $R; mkdir -p $R
python devtools/tools_leaf_stamps.py 6144 > $R/stamps_6144.txt 2>&1; grep -A7 "hr3 start" $R/stamps_6144.txt | head -8
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
python devtools/tools_perf.py 8192x8192x256 4096x4096x64 8192x8192x256 4096x4096x64 2>/dev/null | cut -c1-120
