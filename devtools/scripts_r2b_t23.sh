#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-14s %7dx%-6d nb %3d %7.3f ms %6.2f TF panel %.2f' % ('$name', d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run fold128
run fold256 MI355XQR_TFOLD_MAX=256
run fold128
run fold256 MI355XQR_TFOLD_MAX=256
