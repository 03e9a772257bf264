#!/bin/bash
# round 6, run 3: deferred T merge (Gram + merge tree of a one-launch panel on the update stream in the chain-bound phase): parity + A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run3; mkdir -p $O
python -m pytest tests/test_gpu_multipanel_golden.py tests/test_gpu_qr.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
export CUDA_QR_AMD_LIB=lab
S="16384x16384x256 16384x16384x256 12288x12288x256 8192x8192x256 6144x6144x256"
( for i in 1 2; do
  echo "== MI355XQR_DEFER_T=0"; MI355XQR_DEFER_T=0 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== MI355XQR_DEFER_T=1"; MI355XQR_DEFER_T=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/defer_ab.txt 2>&1
cat $O/defer_ab.txt
python3 devtools/tools_gantt.py 16384x16384x256 > $O/gantt.txt 2>&1; head -3 $O/gantt.txt
