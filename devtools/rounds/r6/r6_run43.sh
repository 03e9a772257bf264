#!/bin/bash
# round 6, run 43: the one-launch panel with up to 128 row workgroups (PF_MAXWG 128): parity of the kernel tests, then the launch on 8192-32768 rows
# (a record: PF_MAXWG 128, MI355XQR_PF_MAX128 and PF_TALL=2 existed only in the tree of this experiment -- profiles/NOTES.md, profiles/r06_panel_fused_128_workgroups_negative.txt)
# (128 x 128 rows against 64 x 256 rows up to 16384; 256-row workgroups beyond), then whole factorisations of 16385-32768 rows against the full-width route
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run43; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_panel_fused.py -m gpu -x -q > $O/tests_pf.log 2>&1; rc=$?; echo "panel tests rc=$rc"; tail -3 $O/tests_pf.log
[ $rc -ne 0 ] && exit 1
export CUDA_QR_AMD_LIB=lab
( echo "== 64 x 256-row (default rule)"; MI355XQR_PF_MAX_ROWS=32768 PF_TALL=2 PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids
  echo "== up to 128 x 128-row (MI355XQR_PF_MAX128=128)"; MI355XQR_PF_MAX128=128 MI355XQR_PF_MAX_ROWS=32768 PF_TALL=1 PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids ) > $O/pf_128wg_perf.txt
cat $O/pf_128wg_perf.txt
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"; }
S="32768x512x128 32768x512x256 24576x512x128 24576x512x256 20480x1024x128 20480x1024x256 32768x2048x128 32768x2048x256 20480x4096x128 24576x4096x128 24576x4096x256 32768x4096x128 32768x4096x256 32768x256x128 32768x256x256"
( for i in 1 2; do
  echo "== default (full-width route above 16384 rows)"; CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== MI355XQR_PF_MAX_ROWS=32768 MI355XQR_CQR_MIN_ROWS=32769"; CHECK=1 MI355XQR_PF_MAX_ROWS=32768 MI355XQR_CQR_MIN_ROWS=32769 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/pf_32768_ab.txt 2>&1
cat $O/pf_32768_ab.txt
