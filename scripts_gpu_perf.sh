#!/bin/bash
run() { echo "== $*"; env "$@" timeout 600 python tools_perf.py 16384x16384x256 16384x16384x128 8192x8192x128 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:300]); continue
    print(d['m'],d['n'],d['nb'],'ms %.1f'%d['ms'],'TF %.2f'%d['tflops'], {k:(round(v['ms'],1),v['tflops']) for k,v in d.items() if isinstance(v,dict)})
"; }
run MI355XQR_PANEL_CUS=32
run MI355XQR_PANEL_CUS=48
run MI355XQR_PANEL_CUS=64
run MI355XQR_PANEL_CUS=80
