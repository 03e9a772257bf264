#!/bin/bash
# round 6, run 5: the one-launch panel with 128-row workgroups (RT = 2): parity, perf table against the 256-row form, and its effect on C3 / C2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_panel_fused.py -m gpu -x -q > $O/tests_fused.log 2>&1; rc=$?; echo "fused tests rc=$rc"; tail -3 $O/tests_fused.log
[ $rc -ne 0 ] && exit 1
( for r in 256 128; do PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py $r 2>&1 | grep -v amdgpu.ids; done ) > $O/panel_fused_perf.txt
cat $O/panel_fused_perf.txt
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
export CUDA_QR_AMD_LIB=lab
S="16384x16384x256 16384x16384x256 8192x8192x256 4096x4096x64 4096x4096x128 4096x4096x256 2048x2048x128 4096x512x128"
( for i in 1 2; do
  echo "== MI355XQR_PF_ROWS=256"; MI355XQR_PF_ROWS=256 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== default (128 where it fits)"; python3 devtools/tools_perf.py $S 2>&1 | fmt
  done
  echo "== default, MI355XQR_FUSED_MIN_ROWS=1024"; MI355XQR_FUSED_MIN_ROWS=1024 python3 devtools/tools_perf.py $S 2>&1 | fmt
) > $O/rows_ab.txt 2>&1
cat $O/rows_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py -m gpu -x -q > $O/tests_qr.log 2>&1; echo "qr tests rc=$?"; tail -3 $O/tests_qr.log
