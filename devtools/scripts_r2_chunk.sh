#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-20s %7.2f ms %6.2f TF panel %.1f  nn %s tn %s' % ('$name', d['ms'], d['tflops'], d.get('panel',{}).get('ms',0), d.get('update_nn'), d.get('vta_tn')))
"; }
run base
run chunk60 MI355XQR_CHUNK_MB=60
run chunk100 MI355XQR_CHUNK_MB=100
run chunk140 MI355XQR_CHUNK_MB=140
run chunk200 MI355XQR_CHUNK_MB=200
run base2
