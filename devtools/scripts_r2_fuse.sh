#!/bin/bash
# A/B of a fusion knob (edit the variable name): tests first, then timings
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_qr.py tests/test_gpu_multipanel_golden.py -x -q -m gpu 2>&1 | tail -3
for c in 0 1; do MI355XQR_FUSE_GRAM=$c timeout 600 python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x128 2048x2048x128 8192x1024x128 262144x512x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('fgram=$c %6dx%-6d nb %3d  %8.3f ms  %6.2f TF  panel %.2f ms' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; done
