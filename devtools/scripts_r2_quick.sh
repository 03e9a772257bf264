#!/bin/bash
# quick timing set (environment passes through):  scripts_r2_quick.sh [tag]
T=${1:-run}
timeout 600 python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x128 2048x2048x128 8192x1024x128 262144x512x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$T %6dx%-6d nb %3d  %8.3f ms  %6.2f TF  panel %.2f ms' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"
