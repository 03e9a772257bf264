#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 4096x4096x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %7dx%-6d nb %3d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run default
run now8 MI355XQR_SMALLT_W8=0
run tn22 MI355XQR_TN_TALL_TILE=22
run default_again
