#!/bin/bash
cd $GRAFT_REPO_ROOT
S="16384x2048x128 32768x2048x128 12288x4096x128 24576x2048x128"
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
echo "== MI355XQR_CQR_MIN_ROWS=196608"; MI355XQR_CQR_MIN_ROWS=196608 python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== MI355XQR_CQR_MIN_ROWS=8193"; MI355XQR_CQR_MIN_ROWS=8193 python3 devtools/tools_perf.py $S 2>&1 | fmt
