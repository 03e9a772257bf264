#!/bin/bash
# A/B of the one-launch panel inside whole factorisations (run from the repo root on the GPU box)
for f in "1 3072" "1 0" "0 0"; do
  set -- $f
  echo "== MI355XQR_FUSED_PANEL=$1 MI355XQR_FUSED_MIN_ROWS=$2"
  MI355XQR_FUSED_PANEL=$1 MI355XQR_FUSED_MIN_ROWS=$2 CHECK=1 python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 4096x512x128 6144x6144x256 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.2f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"
done
