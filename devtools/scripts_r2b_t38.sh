#!/bin/bash
R=gpurun_out/s2v; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 262144x512x128 262144x256x128 131072x256x128 65536x256x128 2097152x512x128 32768x512x128 2>/dev/null | python -c "
import sys, json
print('%-8s' % '$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run tall
run dual MI355XQR_TN_TALL=0
run tall
run dual MI355XQR_TN_TALL=0
CHECK=1 python devtools/tools_perf.py 262144x512x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin: d=json.loads(l); print('resid', d.get('resid'))"
