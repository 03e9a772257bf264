#!/bin/bash
# trailing update kernel: 1 = gemm_nt_kernel (two 8-wave workgroups per CU, one tile each), 2 = gemm_ntp_kernel (two persistent 4-wave workgroups, C prefetch); in situ
O=gpurun_out/ntil; mkdir -p $O; : > $O/ab.txt
for v in 2 1 2 1; do
  echo "== MI355XQR_NT_IL=$v" >> $O/ab.txt
  MI355XQR_NT_IL=$v CHECK=1 timeout -k 10 300 python3 devtools/tools_perf.py 16384x16384x256 8192x8192x256 12288x12288x256 2>&1 | grep -v amdgpu.ids | cut -c1-520 >> $O/ab.txt || exit 1
done
python3 - <<'P'
import json
for l in open('gpurun_out/ntil/ab.txt'):
    l=l.strip()
    if l.startswith('=='): print(l); continue
    d=json.loads(l); print('  %dx%d nb %d %.2f ms  nn %.1f TF/s (%.1f ms)  tn %.1f TF/s (%.1f ms) resid %.1e'%(d['m'],d['n'],d['nb'],d['ms'],d['update_nn']['tflops'],d['update_nn']['ms'],d['vta_tn']['tflops'],d['vta_tn']['ms'],d['resid'][0]))
P
