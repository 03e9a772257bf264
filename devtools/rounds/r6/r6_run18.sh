#!/bin/bash
# round 6, run 18: where does the look-ahead schedule start to pay?  (the rule: n >= 2048, m*n >= 18M, or >= 16M at nb >= 128)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run18; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S="4096x4096x64 4096x4096x128 4096x4096x256 3072x3072x64 3072x3072x128 3072x3072x256 2048x2048x64 2048x2048x128 2048x2048x256 6144x6144x64 6144x6144x128 8192x8192x64 8192x8192x128 8192x2048x128 8192x2048x256 16384x2048x256"
( for la in 0 1 0 1; do
  echo "== MI355XQR_LOOKAHEAD=$la"; MI355XQR_LOOKAHEAD=$la python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/lookahead_rule.txt 2>&1
cat $O/lookahead_rule.txt
