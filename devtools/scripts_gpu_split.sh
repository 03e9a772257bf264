#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python devtools/tools_perf.py $SHAPES 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:300]); continue
    print(d['m'],d['n'],d['nb'],'ms %.1f'%d['ms'],'TF %.2f'%d['tflops'], d.get('resid'))
"; }
export CHECK=1
SHAPES="16384x16384x256 16384x16384x128 8192x8192x256 4096x4096x128"
run MI355XQR_SPLIT=64
run MI355XQR_SPLIT=64:0.45,0
run MI355XQR_SPLIT=64:0.3,0
run MI355XQR_SPLIT=64:0.2,0
run MI355XQR_SPLIT=64:0.6,0
