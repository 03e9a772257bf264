#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 2304x2304x256 2560x2560x256 3072x3072x256 3584x3584x256 4096x4096x256 5120x5120x256 6144x6144x256 4096x3072x256 6144x3072x128 8192x3072x128 12288x3072x128 2>/dev/null | python -c "
import sys, json
print('$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run la_on MI355XQR_LOOKAHEAD=1
run la_off MI355XQR_LOOKAHEAD=0
run la_on MI355XQR_LOOKAHEAD=1
run la_off MI355XQR_LOOKAHEAD=0
