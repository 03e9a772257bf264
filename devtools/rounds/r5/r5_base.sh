#!/bin/bash
# round-5 baseline: tall shapes with and without the two-stream schedule, CQR crossover
cd $GRAFT_REPO_ROOT
S="262144x512x128 131072x256x128 65536x256x128 65536x256x64 4096x512x128"
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
echo "== default"; python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== MI355XQR_LOOKAHEAD=1"; MI355XQR_LOOKAHEAD=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== MI355XQR_CQR_MIN_ROWS=32768"; MI355XQR_CQR_MIN_ROWS=32768 python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== MI355XQR_CQR_MIN_ROWS=32768 MI355XQR_LOOKAHEAD=1"; MI355XQR_CQR_MIN_ROWS=32768 MI355XQR_LOOKAHEAD=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== square"; python3 devtools/tools_perf.py 16384x16384x256 4096x4096x64 8192x8192x256 2>&1 | fmt
