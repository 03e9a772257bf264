#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 262144x512x128 262144x256x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %6dx%-4d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run base
run kmax512 MI355XQR_TN_KMAX=512
run kmax1024 MI355XQR_TN_KMAX=1024
run tn22 MI355XQR_TN_TALL_TILE=22
run tn22_k1024 MI355XQR_TN_TALL_TILE=22 MI355XQR_TN_KMAX=1024
run nn22 MI355XQR_NN_TALL_TILE=22 MI355XQR_SMALLT_W8=0
run now8 MI355XQR_SMALLT_W8=0
run all22 MI355XQR_NN_TALL_TILE=22 MI355XQR_SMALLT_W8=0 MI355XQR_TN_TALL_TILE=22 MI355XQR_TN_KMAX=1024
run k512_nn22 MI355XQR_NN_TALL_TILE=22 MI355XQR_SMALLT_W8=0 MI355XQR_TN_KMAX=512
run nolook MI355XQR_LOOKAHEAD=0
