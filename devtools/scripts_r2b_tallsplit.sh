#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 262144x512x128 262144x256x128 2097152x512x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-40s %7dx%-4d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run base
run nofuse MI355XQR_FUSE_NN=0
run la MI355XQR_LOOKAHEAD=1
for c in 64 96 128 160 192; do
run la_cus${c}_bal0 MI355XQR_LOOKAHEAD=1 MI355XQR_PANEL_CUS=$c MI355XQR_BALANCE=0
run la_cus${c}_bal0_nu MI355XQR_LOOKAHEAD=1 MI355XQR_PANEL_CUS=$c MI355XQR_BALANCE=0 MI355XQR_NEXT=update
done
run la_cus128 MI355XQR_LOOKAHEAD=1 MI355XQR_PANEL_CUS=128
run la_cus128_nb256 MI355XQR_LOOKAHEAD=1 MI355XQR_PANEL_CUS=128 MI355XQR_BALANCE=0 MI355XQR_NB=256
