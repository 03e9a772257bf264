#!/bin/bash
R=gpurun_out/s2c; rm -rf $R; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -4 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 8192x8192x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %6dx%-6d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run early
run noearly MI355XQR_EARLY_NEXT=0
run early_bal1 MI355XQR_BALANCE=7.04,51.5,0.62,0.9
run early_bal2 MI355XQR_BALANCE=7.04,51.5,1.0,0.9
run early_bal3 MI355XQR_BALANCE=7.04,48,1.0,0.9
run early_bal4 MI355XQR_BALANCE=7.04,51.5,1.3,0.9
python devtools/tools_gantt.py 16384x16384x256 > $R/gantt_c3.txt 2>&1
head -20 $R/gantt_c3.txt
