#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 2; do
  echo "== isolated MI355XQR_NT_CEIL=$v"; MI355XQR_NT_CEIL=$v python3 devtools/tools_nt_lab.py 16384x16128x256 16384x8192x256 2>&1 | grep -v amdgpu
done
for v in 0 1 2; do
  echo "== in situ MI355XQR_NT_CEIL=$v (results wrong for 1, 2: rate only)"; MI355XQR_NT_CEIL=$v python3 devtools/tools_perf.py 16384x16384x256 2>&1 | grep -v amdgpu | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], {k: (v['ms'], v['tflops']) for k, v in d.items() if isinstance(v, dict)})
"
done
