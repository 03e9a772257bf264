#!/bin/bash
# round 6, run 35: half-tile route gated at 8 M elements: small shapes back where they were, C2 bench line, multipanel + qr tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run35; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"; }
S="4096x4096x64 2048x2048x64 1024x1024x64 2560x2560x64 3584x3584x64 8192x8192x64 4032x4032x128 4160x4160x256 2048x1024x64 2112x2112x128 1984x1984x0 4160x2112x0 16448x16448x256"
( for i in 1 2; do
  echo "== previous commit"; CHECK=1 CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== half tiles from 8 M elements"; CHECK=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/half_tiles_gated.txt 2>&1
cat $O/half_tiles_gated.txt
python3 bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; python3 -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('C2 bench line', d['ms_per_step'], d['value'], d['accuracy'])"
timeout -k 10 900 python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py tests/test_gpu_kernels.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
