#!/bin/bash
# round 6, run 17: the first panel of a look-ahead factorisation on every compute unit (before the CU partition starts): parity + A/B against HEAD's host layer
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run17; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S="16384x16384x256 16384x16384x256 12288x12288x256 8192x8192x256 4096x4096x256"
( for i in 1 2 3; do
  echo "== previous host layer"; CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== first panel on all CUs"; CUDA_QR_AMD_LIB=lab python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/p0_ab.txt 2>&1
cat $O/p0_ab.txt
