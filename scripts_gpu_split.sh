#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python tools_perf.py $SHAPES 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:300]); continue
    print(d['m'],d['n'],d['nb'],'ms %.1f'%d['ms'],'TF %.2f'%d['tflops'], d.get('resid'), {k:(round(v['ms'],1),v['tflops']) for k,v in d.items() if isinstance(v,dict)})
"; }
export CHECK=1
SHAPES="16384x16384x256 16384x16384x128 8192x8192x256 4096x4096x128 3000x2500x128"
run MI355XQR_TAIL_WIDE=0
run MI355XQR_TAIL_WIDE=0.8
run MI355XQR_TAIL_WIDE=1.5
run MI355XQR_TAIL_WIDE=0.4
