#!/bin/bash
R=gpurun_out/s2u; rm -rf $R; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multipanel_golden.py -q -m gpu -x --timeout=600 -k "cholqr or guard or leaf" > $R/t1.log 2>&1; echo "leaf tests rc=$?"; tail -3 $R/t1.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 262144x512x128 262144x256x128 131072x256x128 65536x256x128 2097152x512x128 32768x512x128 2>/dev/null | python -c "
import sys, json
print('%-8s' % '$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run q4
run q2 MI355XQR_TALL_Q=2
run q4
run q2 MI355XQR_TALL_Q=2
