#!/bin/bash
cd $GRAFT_REPO_ROOT
MI355XQR_NT4=1 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_nt" 2>&1 | tail -2
for n in 1; do for v in 0 1 2; do
  echo "== isolated NT4=$n MI355XQR_NT_CEIL=$v"; MI355XQR_NT4=$n MI355XQR_NT_CEIL=$v python3 devtools/tools_nt_lab.py 16384x16128x256 16384x8192x256 2>&1 | grep -v amdgpu
done; done
echo "== isolated main kernel"; python3 devtools/tools_nt_lab.py 16384x16128x256 16384x8192x256 2>&1 | grep -v amdgpu
for n in 0 1 0 1; do
  echo "== in situ MI355XQR_NT4=$n"; MI355XQR_NT4=$n python3 devtools/tools_perf.py 16384x16384x256 2>&1 | grep -v amdgpu | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], {k: (v['ms'], v['tflops']) for k, v in d.items() if isinstance(v, dict)})
"
done
