#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 1536x1536x256 2048x2048x256 3072x3072x256 4096x4096x256 6144x6144x256 8192x8192x256 10240x10240x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
print('$name', ' '.join('%dx%d:%.2f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run default
run split0 MI355XQR_SPLIT=0
run split32 MI355XQR_SPLIT=32
run split64 MI355XQR_SPLIT=64
run split96 MI355XQR_SPLIT=96
run default
