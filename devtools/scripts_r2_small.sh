#!/bin/bash
for c in 0 1; do MI355XQR_COOP=$c python devtools/tools_perf.py 8192x1024x128 2048x1024x128 16384x1536x128 3000x1000x128 1024x1024x128 2048x2048x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('coop=$c %6dx%-6d nb %3d  %8.3f ms  %6.2f TF  panel %.2f ms' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; done
