#!/bin/bash
# round 6, run 37: ragged TN products in one launch (edge tiles guarded inside the fast kernel's grid): off-grid sizes, before / after
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run37; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm" > $O/tests_k.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -3 $O/tests_k.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'resid', d.get('resid'), {k: (v['ms'], v['tflops']) for k, v in d.items() if isinstance(v, dict)})
"; }
S="4096x4096x64 2048x2048x0 6000x6000x0 5000x5000x0 10000x10000x0 8200x2056x0 4100x4100x256 2000x2000x0 3000x1500x0 1000x1000x0"
( for i in 1 2; do
  echo "== previous commit"; CHECK=1 CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== this tree"; CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/offgrid_ab.txt 2>&1
cut -c1-230 $O/offgrid_ab.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
( timeout -k 5 300 python3 devtools/tools_fuzz_parity.py 3 ragged; timeout -k 5 300 python3 devtools/tools_fuzz_parity.py 6 edges6; timeout -k 5 300 python3 devtools/tools_fuzz_parity.py 1 ) 2>&1 | grep -v amdgpu.ids > $O/fuzz_ragged.txt; grep -c " x " $O/fuzz_ragged.txt; grep "^ok\|Error\|assert" $O/fuzz_ragged.txt
