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
run MI355XQR_SPLIT=64
run MI355XQR_SPLIT=64 MI355XQR_BALANCE=13,43,0.9,0.5
run MI355XQR_SPLIT=64 MI355XQR_BALANCE=13,43,0.6,0.4
run MI355XQR_SPLIT=32 MI355XQR_BALANCE=6,50,1.0,0.6
run MI355XQR_SPLIT=32 MI355XQR_BALANCE=0
run MI355XQR_SPLIT=32 MI355XQR_BALANCE=6,50,1.5,1.0
run MI355XQR_SPLIT=96 MI355XQR_BALANCE=20,36,0.8,0.5
run MI355XQR_SPLIT=32:0.5,64 MI355XQR_BALANCE=10,45,0.9,0.5
