#!/bin/bash
python devtools/tools_perf.py 262144x512x64 262144x512x128 262144x512x256 262144x512x512 262144x256x64 262144x256x128 262144x256x256 65536x256x64 65536x256x128 65536x256x256 131072x256x128 131072x256x256 2097152x512x128 2097152x512x256 32768x2048x128 32768x2048x256 65536x1024x128 65536x1024x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%8dx%-6d nb %3d %8.3f ms %6.2f TF panel %.2f' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 65536x512x128 32768x512x128 131072x1024x128 65536x2048x128 2>/dev/null | python -c "
import sys, json
print('$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run la_default
run la_on MI355XQR_LOOKAHEAD=1
run la_off MI355XQR_LOOKAHEAD=0
