#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-44s %7.2f ms %6.2f TF panel %.1f  nn %s' % ('$name', d['ms'], d['tflops'], d.get('panel',{}).get('ms',0), d.get('update_nn',{}).get('tflops')))
"; }
run base_upd
run auto_tc_0.8_1.4 MI355XQR_NEXT=auto MI355XQR_BALANCE=7,51.5,0.8,1.4
run auto_tc_1.0_1.6 MI355XQR_NEXT=auto MI355XQR_BALANCE=7,51.5,1.0,1.6
run auto_tc_1.2_1.8 MI355XQR_NEXT=auto MI355XQR_BALANCE=7,51.5,1.2,1.8
run panel_tc_1.0_1.6 MI355XQR_NEXT=panel MI355XQR_BALANCE=7,51.5,1.0,1.6
run upd_tc_0.8_1.0 MI355XQR_BALANCE=7,51.5,0.8,1.0
run upd_tc_1.0_1.2 MI355XQR_BALANCE=7,51.5,1.0,1.2
run upd_tc_1.4_0.8 MI355XQR_BALANCE=7,51.5,1.4,0.8
run upd_rp6 MI355XQR_BALANCE=6,51.5,1.1,0.6
run upd_rp8 MI355XQR_BALANCE=8,51.5,1.1,0.6
run upd_ru48 MI355XQR_BALANCE=7,48,1.1,0.6
run base_upd2
