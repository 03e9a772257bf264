#!/bin/bash
R=gpurun_out/s2l; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 4096x4096x64 4096x4096x128 8192x8192x128 262144x512x128 2048x2048x64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-14s %7dx%-6d nb %3d %7.3f ms %6.2f TF panel %.2f' % ('$name', d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run fold128
run nofold MI355XQR_TFOLD_MAX=0
run fold256 MI355XQR_TFOLD_MAX=256
run fold64 MI355XQR_TFOLD_MAX=64
run fold128
