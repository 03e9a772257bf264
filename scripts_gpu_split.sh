#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python tools_perf.py $SHAPES 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:300]); continue
    print(d['m'],d['n'],d['nb'],'ms %.1f'%d['ms'],'TF %.2f'%d['tflops'], d.get('resid'))
"; }
SHAPES="16384x16384x128 16384x16384x256 8192x8192x128"
export CHECK=3
run MI355XQR_SPLIT=64
run MI355XQR_SPLIT=0
run MI355XQR_SPLIT=32:0.6,64
run MI355XQR_LOOKAHEAD=0
