#!/bin/bash
# round 6, run 34: the four-workgroup update kernel on half tiles (M, N = 64 mod 128): every other step of a factorisation at nb 64 used the generic kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run34; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm_nt" > $O/tests_k.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -3 $O/tests_k.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"; }
S="4096x4096x64 2048x2048x64 1024x1024x64 2560x2560x64 4096x4096x32 8192x8192x64 16384x16384x64 4032x4032x128 4160x4160x256 2048x1024x64"
( for i in 1 2; do
  echo "== previous (libmi355xqr_exp_prev.so if present)"; [ -f cuda-qr_amd/libmi355xqr_exp_prev.so ] && CHECK=1 CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== half tiles on the four-workgroup kernel"; CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/half_tiles.txt 2>&1
cat $O/half_tiles.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
