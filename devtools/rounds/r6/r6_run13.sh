#!/bin/bash
# round 6, run 13: look-ahead update behind a deferred merge without the merged T (qrd_trsm_gt): parity + A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run13; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multipanel_golden.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)}, d.get('resid'))
"; }
export CUDA_QR_AMD_LIB=lab
S="16384x16384x256 16384x16384x256 12288x12288x256 8192x8192x256 6144x6144x256 4096x4096x256 4096x4096x128"
( for i in 1 2; do
  echo "== MI355XQR_TRSM=0"; MI355XQR_TRSM=0 CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== MI355XQR_TRSM=1"; MI355XQR_TRSM=1 CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/trsm_ab.txt 2>&1
cat $O/trsm_ab.txt
