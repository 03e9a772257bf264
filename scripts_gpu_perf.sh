#!/bin/bash
mkdir -p gpurun_out
for pc in 0 16 32 64; do echo "PANEL_CUS=$pc"; MI355XQR_PANEL_CUS=$pc timeout 600 python tools_perf.py 16384x16384x128 8192x8192x128 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(d['m'],d['n'],d['nb'],'ms %.1f'%d['ms'],'TF %.2f'%d['tflops'], {k:(round(v['ms'],1),v['tflops']) for k,v in d.items() if isinstance(v,dict)})
"; done
