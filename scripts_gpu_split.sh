#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python tools_perf.py $SHAPES 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:300]); continue
    print(d['m'],d['n'],d['nb'],'ms %.1f'%d['ms'],'TF %.2f'%d['tflops'], {k:(round(v['ms'],1),v['tflops']) for k,v in d.items() if isinstance(v,dict)})
"; }
SHAPES="16384x16384x256 16384x16384x128 8192x8192x256"
run MI355XQR_BALANCE=13,43,1.5,1.2
run MI355XQR_BALANCE=14,44,0.7,0.4
run MI355XQR_BALANCE=14,44,0.9,0.5
run MI355XQR_BALANCE=14,44,1.1,0.6
run MI355XQR_BALANCE=14,44,1.3,0.8
run MI355XQR_BALANCE=12,44,1.0,0.5
run MI355XQR_BALANCE=16,44,1.0,0.5
