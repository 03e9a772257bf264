#!/bin/bash
R=gpurun_out/s2i; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -4 $R/tests.log
python devtools/tools_perf.py 16384x16384x256 16384x16384x256 12288x12288x256 8192x8192x256 2>/dev/null | cut -c1-330
