#!/bin/bash
# issue-order variants of the 128 x 128 TN products' K loop, in situ
O=gpurun_out/kpipe; mkdir -p $O; : > $O/ab.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_multipanel_golden.py -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -1 $O/pytest.txt
for v in 3 0 3 0; do
  echo "== MI355XQR_KPIPE=$v" >> $O/ab.txt
  MI355XQR_KPIPE=$v CHECK=1 timeout -k 10 300 python3 devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 262144x512x128 2>&1 | grep -v amdgpu.ids | cut -c1-620 >> $O/ab.txt || exit 1
done
python3 - <<'P'
import json
for l in open('gpurun_out/kpipe/ab.txt'):
    l=l.strip()
    if l.startswith('=='): print(l); continue
    d=json.loads(l); print('  %dx%d %.2f ms  '%(d['m'],d['n'],d['ms']) + ' '.join('%s %.1f ms'%(k,v['ms']) for k,v in d.items() if isinstance(v,dict)) + ' resid %.1e'%d['resid'][0])
P
