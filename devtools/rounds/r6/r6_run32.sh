#!/bin/bash
# round 6, run 32: the share of the wide update the panel stream takes (balance_cols) with a chain time of its own (lab knob MI355XQR_BALANCE_E=tc0,tc1),
# on the 64-CU panel stream whose tall panels are one launch now (Gantt of run 31: that stream idles 0.4-0.85 ms per step)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run32; mkdir -p $O
export CUDA_QR_AMD_LIB=lab
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S="16384x8192x256 16384x6144x256 12288x4096x256 16384x4096x256 9216x9216x256"
( for e in default 1.0,0.3 0.8,0.2 0.6,0.1 0.4,0.0 0.2,0.0; do
  echo "== MI355XQR_BALANCE_E=$e (default split)"
  if [ $e = default ]; then python3 devtools/tools_perf.py $S 2>&1 | fmt; else MI355XQR_BALANCE_E=$e python3 devtools/tools_perf.py $S 2>&1 | fmt; fi
  done
  S3="16384x16384x256 12288x12288x256 10240x10240x256"
  for e in default 1.0,0.3 0.8,0.2 0.6,0.1 0.4,0.0; do
  echo "== MI355XQR_SPLIT=64 MI355XQR_BALANCE_E=$e"
  if [ $e = default ]; then MI355XQR_SPLIT=64 python3 devtools/tools_perf.py $S3 2>&1 | fmt; else MI355XQR_SPLIT=64 MI355XQR_BALANCE_E=$e python3 devtools/tools_perf.py $S3 2>&1 | fmt; fi
  done
  echo "== split 32 (default)"; python3 devtools/tools_perf.py $S3 2>&1 | fmt
  for e in 1.4,0.5 0.9,0.5; do echo "== split 32 MI355XQR_BALANCE_E=$e"; MI355XQR_BALANCE_E=$e python3 devtools/tools_perf.py $S3 2>&1 | fmt; done
  ) > $O/balance_e.txt 2>&1
cat $O/balance_e.txt
