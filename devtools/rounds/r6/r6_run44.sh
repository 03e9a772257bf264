#!/bin/bash
# round 6, run 44: odd sizes (odd leading dimension: no 16-byte aligned columns) against their even neighbours, device API and qr_device (host pointers)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run44; mkdir -p $O
python3 devtools/tools_perf.py 5000x5000x0 5001x5001x0 5002x5002x0 8191x8191x0 8192x8192x0 10001x3001x0 10000x3000x0 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], {k: (v['ms'], v['tflops']) for k, v in d.items() if isinstance(v, dict)})
" > $O/odd.txt; cut -c1-200 $O/odd.txt
for s in "5000 5000" "5001 5001" "8191 8191" "8192 8192"; do timeout -k 5 60 ./cuda-qr_amd/build/qr_device $s 2>&1 | grep "MMQR ran"; done | tee -a $O/odd.txt
