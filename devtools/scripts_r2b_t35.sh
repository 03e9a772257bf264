#!/bin/bash
R=gpurun_out/s2t; rm -rf $R; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -2 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 8192x8192x256 2>/dev/null | python -c "
import sys, json
print('%-10s' % '$name', ' '.join('%dx%d:%.2f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run vt_panel
run vt_update MI355XQR_VT_ON_PANEL=0
run vt_panel
run vt_update MI355XQR_VT_ON_PANEL=0
