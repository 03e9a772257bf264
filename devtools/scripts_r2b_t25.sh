#!/bin/bash
python devtools/tools_perf.py 32768x8192x128 32768x8192x256 16384x8192x128 16384x8192x256 12288x8192x128 12288x8192x256 8192x6144x128 8192x6144x256 4096x3072x128 4096x3072x256 6144x4096x128 6144x4096x256 1536x1536x128 1536x1536x256 1024x1024x128 1024x1024x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%7dx%-6d nb %3d %7.3f ms %6.2f TF panel %.2f' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"
