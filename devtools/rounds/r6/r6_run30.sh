#!/bin/bash
# round 6, run 30: one-launch panels up to 16384 rows as the default: full GPU suite, what the library now picks by itself (MxNx0), and the
# balance model's chain time on the 64-CU panel stream now that its tall panels are one launch (lab knob MI355XQR_BALANCE=Rp,Ru,tc0,tc1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run30; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S=""
for s in 10240x512 16384x512 10240x1024 16384x1024 10240x2048 12288x2048 16384x2048 10240x3072 12288x3072 16384x3072 10240x4096 12288x4096 16384x4096 20480x4096 16384x6144 16384x8192 12288x12288 16384x16384 8192x8192 9216x9216 32768x2048; do S="$S ${s}x0"; done
( for i in 1 2; do python3 devtools/tools_perf.py $S 2>&1 | fmt; done ) > $O/default_rule.txt 2>&1
cat $O/default_rule.txt
export CUDA_QR_AMD_LIB=lab
S2="16384x8192x0 12288x4096x0 16384x4096x0 12288x3072x0 16384x6144x0 12288x2048x0 9216x9216x0"
( for b in default 14.08,44.16,1.1,0.6 14.08,44.16,0.7,0.3 14.08,44.16,0.5,0.15 14.08,44.16,0.35,0.1; do
  echo "== MI355XQR_BALANCE=$b"
  if [ $b = default ]; then python3 devtools/tools_perf.py $S2 2>&1 | fmt; else MI355XQR_BALANCE=$b python3 devtools/tools_perf.py $S2 2>&1 | fmt; fi
  done ) > $O/balance64.txt 2>&1
cat $O/balance64.txt
