#!/bin/bash
R=gpurun_out/s2p; rm -rf $R; mkdir -p $R
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 8192x8192x256 4096x4096x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-20s %7dx%-6d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run default
run hop0 MI355XQR_HOP_FROM=0
run hop2 MI355XQR_HOP_FROM=2
run hop4 MI355XQR_HOP_FROM=4
run hop0_leaf20 MI355XQR_HOP_FROM=0 MI355XQR_HOP_LEAF_MS=0.2
run default
MI355XQR_HOP_FROM=0 timeout -k 10 300 python -m pytest tests/test_gpu_multipanel_golden.py -q -m gpu -x --timeout=300 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
