#!/bin/bash
# the stacked panels of the TSQR tree (P x 128 rows): the leaf chain against the one-launch panel at small heights
R=gpurun_out/r5_stacked; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for fm in 3072 1024 512; do
  export MI355XQR_FUSED_MIN_ROWS=$fm
  echo "MI355XQR_FUSED_MIN_ROWS=$fm"
  timeout -k 10 200 python3 devtools/tools_perf.py 512x128x128 1024x128x128 2048x128x128 1024x512x128 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(d['m'], d['n'], 'ms %.3f' % d['ms'])
"
  timeout -k 10 200 python3 devtools/tools_tsqr_latency.py 262144x512x8x128 65536x256x4x128 131072x256x2x128 2>&1 | grep -v amdgpu.ids | cut -c1-220
done
