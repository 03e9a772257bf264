#!/bin/bash
# round 6, run 40: ragged NN products -- all-guarded only below 60 MFLOP (was 400), interior + strips above; previous commit against this tree, no CHECK
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run40; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], {k: (v['ms'], v['tflops']) for k, v in d.items() if isinstance(v, dict)})
"; }
S="4096x4096x64 2048x2048x0 2000x2000x0 3000x1500x0 1000x1000x0 1500x1500x0 6000x6000x0 5000x5000x0 4100x4100x256 8200x2056x0 10000x10000x0 8200x8200x256 3000x3000x0 16384x16384x256"
( for i in 1 2; do
  echo "== previous commit"; CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== this tree"; python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/nn_ragged_ab.txt 2>&1
cut -c1-200 $O/nn_ragged_ab.txt
