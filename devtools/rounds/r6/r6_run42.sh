#!/bin/bash
# round 6, run 42: look-ahead at nb 64 again, now that the update kernel takes the half tiles of every other step (C2: 4096^2 at nb 64)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run42; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S="4096x4096x64 3072x3072x64 6144x6144x64 8192x8192x64 4096x2048x64 8192x2048x64"
( for la in 0 1 0 1; do for sp in 64 32; do
  [ $la = 0 ] && [ $sp = 32 ] && continue
  echo "== MI355XQR_LOOKAHEAD=$la MI355XQR_SPLIT=$sp"; MI355XQR_LOOKAHEAD=$la MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py $S 2>&1 | fmt
  done; done ) > $O/la_nb64.txt 2>&1
cat $O/la_nb64.txt
