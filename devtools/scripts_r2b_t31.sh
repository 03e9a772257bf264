#!/bin/bash
R=gpurun_out/s2r; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 2048x2048x256 262144x512x128 262144x256x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%8dx%-6d nb %3d %8.3f ms %6.2f TF' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops']))
"
