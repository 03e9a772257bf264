#!/bin/bash
R=gpurun_out/s2s; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 8192x8192x256 6144x6144x256 5120x5120x256 2>/dev/null | python -c "
import sys, json
print('%-10s' % '$name', ' '.join('%dx%d:%.2f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run split_t
run no_split MI355XQR_SPLIT_T=0
run split_t
run no_split MI355XQR_SPLIT_T=0
