#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 262144x512x128 262144x256x128 131072x256x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-26s %7dx%-4d %7.3f ms %6.2f TF panel %.2f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run base
for sl in 2 4 16; do run slots$sl MI355XQR_DUAL_SLOTS=$sl; done
for km in 8 16 32 64; do run kmin$km MI355XQR_DUAL_KMIN=$km; done
run slots4_kmin16 MI355XQR_DUAL_SLOTS=4 MI355XQR_DUAL_KMIN=16
run slots2_kmin32 MI355XQR_DUAL_SLOTS=2 MI355XQR_DUAL_KMIN=32
run kmax512 MI355XQR_TN_KMAX=512
run rb1 MI355XQR_TALL_RB=1
run rb2 MI355XQR_TALL_RB=2
run nb256 MI355XQR_NB=256
run gy0 MI355XQR_FUSE_NN_GY=0
run base_again
