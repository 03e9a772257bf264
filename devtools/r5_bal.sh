#!/bin/bash
cd $GRAFT_REPO_ROOT
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'])
"; }
for b in "7.04,51.5,1.1,0.6" "7.04,51.5,1.2,0.7" "7.04,51.5,1.3,0.7" "7.04,51.5,1.4,0.8" "7.04,51.5,1.6,0.9" "6.5,51.5,1.1,0.6" "6.0,52.5,1.2,0.7" "7.04,50,1.1,0.6"; do
  echo "== MI355XQR_BALANCE=$b"; MI355XQR_BALANCE="$b" python3 devtools/tools_perf.py 16384x16384x256 12288x12288x256 2>&1 | fmt
done
