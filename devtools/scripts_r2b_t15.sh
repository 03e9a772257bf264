#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 8192x8192x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %7dx%-6d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run default
run hwq8 GPU_MAX_HW_QUEUES=8
run split30 MI355XQR_SPLIT=32:0.3,64
run split30_hwq8 MI355XQR_SPLIT=32:0.3,64 GPU_MAX_HW_QUEUES=8
run split40_hwq8 MI355XQR_SPLIT=32:0.4,64 GPU_MAX_HW_QUEUES=8
run split30_hwq16 MI355XQR_SPLIT=32:0.3,64 GPU_MAX_HW_QUEUES=16
run default
