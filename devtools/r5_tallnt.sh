#!/bin/bash
cd $GRAFT_REPO_ROOT
S="262144x512x128 131072x256x128 65536x256x128 524288x512x128 2097152x512x128 32768x2048x128"
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)}, d.get('resid'))
"; }
echo "== MI355XQR_TALL_NT=0"; MI355XQR_TALL_NT=0 CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== MI355XQR_TALL_NT=1"; MI355XQR_TALL_NT=1 CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
