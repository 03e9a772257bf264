#!/bin/bash
# round 6, run 6: preconditioned retry of refused tall panels (parity, price), Q_top prefetch, row-split rule, FUSED_MIN_ROWS sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run6; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_panel_cqr.py tests/test_gpu_panel_fused.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log
( python3 devtools/tools_cond.py 0 262144x512x128 65536x256x128; python3 devtools/tools_cond.py 1e9 262144x512x128 65536x256x128 ) 2>&1 | grep -v amdgpu.ids > $O/price.txt
CUDA_QR_AMD_LIB=lab MI355XQR_CQR_RETRY=0 python3 devtools/tools_cond.py 1e9 262144x512x128 65536x256x128 2>&1 | grep -v amdgpu.ids | sed 's/^/retry off: /' >> $O/price.txt
cat $O/price.txt
PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids > $O/panel_fused_perf_rule.txt; cat $O/panel_fused_perf_rule.txt
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
S="16384x16384x256 16384x16384x256 8192x8192x256 4096x4096x64 4096x4096x256 2048x2048x128 1024x1024x128 4096x512x128 2048x512x128"
( for r in 3072 2048 1024 512 256 3072; do echo "== MI355XQR_FUSED_MIN_ROWS=$r"; MI355XQR_FUSED_MIN_ROWS=$r python3 devtools/tools_perf.py $S 2>&1 | fmt; done ) > $O/fused_min_rows.txt 2>&1
cat $O/fused_min_rows.txt
