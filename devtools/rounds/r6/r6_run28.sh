#!/bin/bash
# round 6, run 28: the one-launch panel beyond 8192 rows (64 row workgroups of 256 rows: up to 16384 rows where the stream has 65 compute units)?
# lab knob MI355XQR_PF_MAX_ROWS; kernel times, then whole factorisations with residual checks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run28; mkdir -p $O
export CUDA_QR_AMD_LIB=lab
( MI355XQR_PF_MAX_ROWS=16384 PF_TALL=1 PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids ) > $O/pf_tall_perf.txt; cat $O/pf_tall_perf.txt
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"; }
S="12288x4096x0 16384x4096x0 16384x8192x0 20480x4096x0 12288x12288x0 16384x16384x256 10240x2048x0 12288x2048x0 16384x2048x0 16384x2048x256 12288x3072x0 12288x3072x256 16384x1024x0 16384x1024x256 12288x512x256 16384x512x256 16384x512x128"
( for i in 1 2; do
  echo "== default"; CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== MI355XQR_PF_MAX_ROWS=16384"; CHECK=1 MI355XQR_PF_MAX_ROWS=16384 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/pf_tall_ab.txt 2>&1
cat $O/pf_tall_ab.txt
