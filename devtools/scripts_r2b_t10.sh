#!/bin/bash
R=gpurun_out/s2k; rm -rf $R; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-20s %7dx%-6d %7.2f ms %6.2f TF panel %.1f tn %s nn %s' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0), d.get('vta_tn',{}).get('tflops'), d.get('update_nn',{}).get('tflops')))
"; }
run w1_2048
run w1_256 MI355XQR_EARLY_W1=256
run w1_1024 MI355XQR_EARLY_W1=1024
run w1_4096 MI355XQR_EARLY_W1=4096
run w1_2048
