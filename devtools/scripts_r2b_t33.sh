#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-26s %7dx%-6d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run default
run pad0_1 MI355XQR_PAIR_PAD0=1
run pad0_2 MI355XQR_PAIR_PAD0=2
run pad0_3 MI355XQR_PAIR_PAD0=3
run split40 MI355XQR_SPLIT=32:0.4,64
run split40_pad1 MI355XQR_SPLIT=32:0.4,64 MI355XQR_PAIR_PAD=1
run split40_pad2 MI355XQR_SPLIT=32:0.4,64 MI355XQR_PAIR_PAD=2
run split40_pad3 MI355XQR_SPLIT=32:0.4,64 MI355XQR_PAIR_PAD=3
run split30_pad2 MI355XQR_SPLIT=32:0.3,64 MI355XQR_PAIR_PAD=2
run split20_pad2 MI355XQR_SPLIT=32:0.2,64 MI355XQR_PAIR_PAD=2
