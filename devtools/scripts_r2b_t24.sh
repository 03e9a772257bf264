#!/bin/bash
python devtools/tools_perf.py 2048x2048x128 2048x2048x256 3072x3072x128 3072x3072x256 4096x4096x128 4096x4096x256 6144x6144x128 6144x6144x256 8192x4096x128 8192x4096x256 16384x2048x128 16384x2048x256 32768x1024x128 32768x1024x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%7dx%-6d nb %3d %7.3f ms %6.2f TF panel %.2f' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"
