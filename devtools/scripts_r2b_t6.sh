#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %7dx%-6d nb %3d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run default
run split_32_64_at50 MI355XQR_SPLIT=32:0.5,64
run split_32_64_at40 MI355XQR_SPLIT=32:0.4,64
run split_32_64_at30 MI355XQR_SPLIT=32:0.3,64
run split_64 MI355XQR_SPLIT=64
run split_32_96_at40 MI355XQR_SPLIT=32:0.4,96
run default_again
