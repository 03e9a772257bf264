#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_qr.py -q -m gpu --timeout=600 -x > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/tests.log
run() { echo "== $*"; env "$@" timeout 600 python devtools/tools_perf.py 16384x16384x256 16384x16384x128 8192x8192x256 4096x4096x128 262144x512x128 65536x256x128 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:300]); continue
    print(d['m'],d['n'],d['nb'],'ms %.1f'%d['ms'],'TF %.2f'%d['tflops'], d.get('resid'), {k:(round(v['ms'],1),v['tflops']) for k,v in d.items() if isinstance(v,dict)})
"; }
export CHECK=1
run MI355XQR_X=1
run MI355XQR_LOOKAHEAD=0
