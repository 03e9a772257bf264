#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
print('%-10s' % '$name', ' '.join('%dx%d:%.2f(tn %.1f)' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms'], json.loads(l).get('vta_tn',{}).get('tflops',0)) for l in sys.stdin))
"; }
for r in 1 2; do
run w1_2048
run w1_4096 MI355XQR_EARLY_W1=4096
run w1_6144 MI355XQR_EARLY_W1=6144
run w1_8192 MI355XQR_EARLY_W1=8192
run w1_1024 MI355XQR_EARLY_W1=1024
done
