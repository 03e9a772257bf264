#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-20s %7dx%-6d %7.2f ms %6.2f TF panel %.1f tn %s nn %s' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0), d.get('vta_tn',{}).get('tflops'), d.get('update_nn',{}).get('tflops')))
"; }
run default32
run split16 MI355XQR_SPLIT=16
run split24 MI355XQR_SPLIT=24
run split40 MI355XQR_SPLIT=40
run split48 MI355XQR_SPLIT=48
run split24_32at50 MI355XQR_SPLIT=24:0.5,32
run default32
