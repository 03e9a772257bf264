#!/bin/bash
# CU split of the chain-bound tail, with the one-launch panel (run from the repo root on the GPU box)
for sp in "" "32:0.36,64" "32:0.36,96" "32:0.36,U" "32:0.30,64" "32:0.45,64" "32:0.36,48"; do
  echo "== MI355XQR_SPLIT=$sp"
  if [ -z "$sp" ]; then unset MI355XQR_SPLIT; else export MI355XQR_SPLIT=$sp; fi
  python devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.2f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"
done
