#!/bin/bash
# round 6, run 11: Gram blocks of narrow panels out of the one-launch panel itself (MI355XQR_FUSED_GRAM = widest panel that does so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run11; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
export CUDA_QR_AMD_LIB=lab
S="4096x4096x64 4096x4096x128 4096x4096x256 2048x2048x64 2048x2048x128 1024x1024x128 4096x512x128 2048x512x128 8192x8192x256 8192x8192x128"
( for g in 0 64 128 256 0 128; do echo "== MI355XQR_FUSED_GRAM=$g"; MI355XQR_FUSED_GRAM=$g python3 devtools/tools_perf.py $S 2>&1 | fmt; done ) > $O/fused_gram.txt 2>&1
cat $O/fused_gram.txt
unset CUDA_QR_AMD_LIB
timeout -k 10 600 python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
