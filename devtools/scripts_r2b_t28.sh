#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 8192x1024x128 16384x1024x128 32768x1024x128 65536x1024x128 8192x2048x128 16384x2048x128 32768x2048x128 65536x2048x128 16384x4096x128 32768x4096x128 65536x4096x128 8192x4096x128 12288x4096x128 2>/dev/null | python -c "
import sys, json
print('$name', ' '.join('%dx%d:%.2f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run la_on MI355XQR_LOOKAHEAD=1
run la_off MI355XQR_LOOKAHEAD=0
run la_on_nosplit MI355XQR_LOOKAHEAD=1 MI355XQR_SPLIT=0
run la_on MI355XQR_LOOKAHEAD=1
run la_off MI355XQR_LOOKAHEAD=0
