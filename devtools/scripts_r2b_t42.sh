#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 8192x8192x256 6144x6144x256 10240x10240x256 5120x5120x256 2>/dev/null | python -c "
import sys, json
print('%-14s' % '$name', ' '.join('%dx%d:%.2f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
for r in 1 2; do
run default
run bal_off MI355XQR_BALANCE=0
run tc_lo MI355XQR_BALANCE=14.08,44.16,0.7,0.6
run tc_hi MI355XQR_BALANCE=14.08,44.16,1.5,0.6
run early_off MI355XQR_EARLY_NEXT=0
done
