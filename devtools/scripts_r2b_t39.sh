#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-22s %7.2f ms tn %s nn %s panel %.1f' % ('$name', d['ms'], d.get('vta_tn',{}).get('tflops'), d.get('update_nn',{}).get('tflops'), d.get('panel',{}).get('ms',0)))
"; }
run default
run gm0 MI355XQR_NT_GM=0
run gm4 MI355XQR_NT_GM=4
run gm16 MI355XQR_NT_GM=16
run gm7 MI355XQR_NT_GM=7
run bal_a MI355XQR_BALANCE=7.04,51.5,1.3,0.6
run bal_b MI355XQR_BALANCE=7.04,51.5,1.6,0.6
run bal_c MI355XQR_BALANCE=7.04,48,1.3,0.6
run bal_d MI355XQR_BALANCE=8,51.5,1.1,0.6
run w1_4096 MI355XQR_EARLY_W1=4096
run default
