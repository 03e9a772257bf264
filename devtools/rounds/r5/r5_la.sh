#!/bin/bash
# two-stream schedule for tall shapes: panel chain on an UNMASKED stream, wide update on a CU-masked one (a few CUs always free for the one-workgroup kernels)
cd $GRAFT_REPO_ROOT
S="262144x512x128 131072x256x128 524288x512x128"
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
echo "== default (single stream)"; python3 devtools/tools_perf.py $S 2>&1 | fmt
for sp in "8:1.0,U" "16:1.0,U" "32:1.0,U" "64:1.0,U"; do
  echo "== LOOKAHEAD=1 SPLIT=$sp"; MI355XQR_LOOKAHEAD=1 MI355XQR_SPLIT="$sp" python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== LOOKAHEAD=1 SPLIT=$sp NEXT=panel"; MI355XQR_LOOKAHEAD=1 MI355XQR_SPLIT="$sp" MI355XQR_NEXT=panel python3 devtools/tools_perf.py $S 2>&1 | fmt
done
