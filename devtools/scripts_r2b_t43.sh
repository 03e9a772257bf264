#!/bin/bash
R=gpurun_out/s2w; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -2 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 8192x8192x256 4096x4096x64 6144x6144x256 16384x16384x256 2048x2048x256 2>/dev/null | python -c "
import sys, json
print('%-6s' % '$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
for r in 1 2; do
run q5
run q3 MI355XQR_LEAF_Q5=0
done
