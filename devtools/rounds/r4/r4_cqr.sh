#!/bin/bash
# the full-width tall panel against the leaf chain on tall shapes (run from the repo root on the GPU box)
for mr in 32768 0; do
  echo "== MI355XQR_CQR_MIN_ROWS=$mr"
  MI355XQR_CQR_MIN_ROWS=$mr CHECK=1 python devtools/tools_perf.py 262144x512x128 131072x256x128 196608x256x128 524288x256x128 1048576x128x128 2097152x128x128 2097152x512x128 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"
done
