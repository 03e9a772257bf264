#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %7dx%-6d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run default
run split25 MI355XQR_SPLIT=32:0.25,64
run split20 MI355XQR_SPLIT=32:0.2,64
run split15 MI355XQR_SPLIT=32:0.15,64
run split10 MI355XQR_SPLIT=32:0.1,64
run split20_96 MI355XQR_SPLIT=32:0.2,96
run split15_128 MI355XQR_SPLIT=32:0.15,128
run default
