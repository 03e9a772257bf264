#!/bin/bash
# round 6, run 7: how low can MI355XQR_FUSED_MIN_ROWS go; look-ahead threshold with the faster one-launch panel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run7; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
S="4096x4096x64 4096x4096x128 4096x4096x256 3072x3072x128 2048x2048x128 1024x1024x128 512x512x128 512x128x128 4096x512x128 2048x512x128 1024x512x128 2048x256x128"
( for r in 256 128 32 256; do echo "== MI355XQR_FUSED_MIN_ROWS=$r"; MI355XQR_FUSED_MIN_ROWS=$r python3 devtools/tools_perf.py $S 2>&1 | fmt; done
  for la in 0 1; do echo "== MI355XQR_FUSED_MIN_ROWS=256 MI355XQR_LOOKAHEAD=$la"; MI355XQR_FUSED_MIN_ROWS=256 MI355XQR_LOOKAHEAD=$la python3 devtools/tools_perf.py 4096x4096x64 4096x4096x128 4096x4096x256 3072x3072x128 3072x3072x256 2048x2048x128 6144x6144x256 2>&1 | fmt; done
) > $O/sweeps.txt 2>&1
cat $O/sweeps.txt
