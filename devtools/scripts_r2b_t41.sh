#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 262144x512x128 262144x256x128 131072x256x128 2097152x512x128 2>/dev/null | python -c "
import sys, json
print('%-8s' % '$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
for r in 1 2; do
run g512
run g256 MI355XQR_TALL_QGRID=256
run g768 MI355XQR_TALL_QGRID=768
run g1024 MI355XQR_TALL_QGRID=1024
done
