#!/bin/bash
for sp in 64 0 32; do MI355XQR_SPLIT=$sp python devtools/tools_perf.py 2048x2048x128 4096x4096x128 8192x8192x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('split=$sp %6dx%-6d nb %3d  %8.3f ms  %6.2f TF  panel %.2f ms' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; done
