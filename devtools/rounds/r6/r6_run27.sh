#!/bin/bash
# round 6, run 27: the full-width 128-column route for the tall panels of a look-ahead plan (panel stream of 32 / 64 CUs)?  nb 128 plans, lab knob
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run27; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"; }
export CUDA_QR_AMD_LIB=lab
S="16384x16384x128 12288x12288x128 16384x8192x128 32768x8192x128"
( for i in 1 2; do
  echo "== default (leaf chain above 8192 rows on look-ahead plans)"; CHECK=1 MI355XQR_LOOKAHEAD=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== MI355XQR_CQR_MIN_ROWS_LA=8193 (poll)"; CHECK=1 MI355XQR_LOOKAHEAD=1 MI355XQR_CQR_MIN_ROWS_LA=8193 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== MI355XQR_CQR_MIN_ROWS_LA=8193 MI355XQR_GUARD=latch"; CHECK=1 MI355XQR_LOOKAHEAD=1 MI355XQR_GUARD=latch MI355XQR_CQR_MIN_ROWS_LA=8193 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/cqr_on_la.txt 2>&1
cat $O/cqr_on_la.txt
