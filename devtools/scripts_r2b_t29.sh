#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 1024x1024x256 1536x1536x256 2048x1024x128 4096x1024x128 4096x512x128 1024x512x128 2048x1536x256 1900x1900x256 2048x2048x256 3072x2048x256 4096x2048x128 2>/dev/null | python -c "
import sys, json
print('$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run la_on MI355XQR_LOOKAHEAD=1
run la_off MI355XQR_LOOKAHEAD=0
run la_on MI355XQR_LOOKAHEAD=1
run la_off MI355XQR_LOOKAHEAD=0
