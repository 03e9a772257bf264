#!/bin/bash
# parked full-width panels (V written once, into A; MI355XQR_CQR_PARK): parity on tall shapes, TSQR, timing A/B
R=gpurun_out/r5_park; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_panel_cqr.py tests/test_gpu_qr.py -x -q -k "cqr or tall or tsqr or c4 or c5 or fuzz or thin" > $R/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $R/tests.txt
timeout -k 5 400 python3 devtools/tools_cqr_fuzz.py > $R/cqr_fuzz.txt 2>&1; tail -2 $R/cqr_fuzz.txt
for pk in 1 0; do
  export MI355XQR_CQR_PARK=$pk
  echo "MI355XQR_CQR_PARK=$pk"
  python3 devtools/tools_perf.py 262144x512x128 131072x256x128 65536x256x128 2097152x512x128 262144x256x128 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(d['m'], d['n'], 'ms %.3f' % d['ms'], 'panel', round(d.get('panel',{}).get('ms',0),2), 'resid', d.get('resid'))
"
  python3 devtools/tools_tsqr_latency.py 262144x512x8x128 65536x256x4x128 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
