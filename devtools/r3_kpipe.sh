#!/bin/bash
# issue-order variants of the wide TN product's K loop, in situ
O=gpurun_out/kpipe; mkdir -p $O; : > $O/ab.txt
for v in 3 0 3 0; do
  echo "== MI355XQR_KPIPE=$v" >> $O/ab.txt
  MI355XQR_KPIPE=$v CHECK=1 timeout -k 10 300 python3 devtools/tools_perf.py 16384x16384x256 8192x8192x256 12288x12288x256 2>&1 | grep -v amdgpu.ids | cut -c1-520 >> $O/ab.txt || exit 1
done
python3 - <<'P'
import json
for l in open('gpurun_out/kpipe/ab.txt'):
    l=l.strip()
    if l.startswith('=='): print(l); continue
    d=json.loads(l); print('  %dx%d %.2f ms  nn %.1f TF/s  tn %.1f TF/s (%.1f ms) resid %.1e'%(d['m'],d['n'],d['ms'],d['update_nn']['tflops'],d['vta_tn']['tflops'],d['vta_tn']['ms'],d['resid'][0]))
P
