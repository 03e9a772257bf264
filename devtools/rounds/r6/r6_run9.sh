#!/bin/bash
# round 6, run 9: two-part hand-off of the leaf factors in the one-launch panel (U'^-1 as soon as the LU is done; T, R behind): parity + A/B
# against the previous kernel (libmi355xqr_exp_prev.so: the lab library with HEAD's qr_panel_fused.hip)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run9; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_panel_fused.py tests/test_gpu_multipanel_golden.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
( for i in 1 2; do
  echo "== previous kernel"; CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids
  echo "== two-part hand-off"; CUDA_QR_AMD_LIB=lab PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids
  done ) > $O/panel_fused_perf_ab.txt
cat $O/panel_fused_perf_ab.txt
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
S="16384x16384x256 16384x16384x256 8192x8192x256 4096x4096x64 4096x4096x256 2048x2048x128 4096x512x128"
( for i in 1 2; do
  echo "== previous kernel"; CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== two-part hand-off"; CUDA_QR_AMD_LIB=lab python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/whole_ab.txt 2>&1
cat $O/whole_ab.txt
