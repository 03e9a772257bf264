#!/bin/bash
# full-width route below 8193 rows after the one-workgroup kernels lost another 60 us (single-stream plans, TSQR stacked panels)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="512x128x128 1024x128x128 2048x128x128 1024x512x128 2048x512x128 3072x256x128 4096x256x128 4096x512x128 6144x384x128 8192x256x128 8192x512x128 8192x1024x128"
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:160]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'])
"; }
for r in 8193 256; do
  echo "== MI355XQR_CQR_MIN_ROWS=$r"; MI355XQR_CQR_MIN_ROWS=$r timeout -k 10 300 python3 devtools/tools_perf.py $S 2>&1 | fmt
  MI355XQR_CQR_MIN_ROWS=$r timeout -k 10 200 python3 devtools/tools_tsqr_latency.py 262144x512x8x128 65536x256x4x128 131072x256x2x128 2>&1 | grep -v amdgpu.ids | cut -c1-220
done
