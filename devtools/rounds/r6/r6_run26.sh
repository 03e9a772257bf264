#!/bin/bash
# round 6, run 26: does a second CU-partition phase work once the runtime has more than four hardware queues (GPU_MAX_HW_QUEUES=8)?
# (r6_run4: with the default four, the second phase's panel and update streams share a queue and serialise); + the new edge-shape test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run26; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_qr.py -m gpu -x -q -k "thresholds or split" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S="16384x16384x256 8192x8192x256 12288x12288x256"
( for q in 4 8 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$q default split"; GPU_MAX_HW_QUEUES=$q python3 devtools/tools_perf.py $S 2>&1 | fmt
  for sp in "32:0.25,64" "32:0.4,64" "32:0.15,64"; do
    echo "== GPU_MAX_HW_QUEUES=$q MI355XQR_SPLIT=$sp"; GPU_MAX_HW_QUEUES=$q MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py $S 2>&1 | fmt
  done; done ) > $O/hwq_split.txt 2>&1
cat $O/hwq_split.txt
GPU_MAX_HW_QUEUES=8 MI355XQR_SPLIT=32:0.25,64 python3 devtools/tools_gantt.py 16384x16384x256 > $O/gantt_hwq8_split.txt 2>&1
tail -36 $O/gantt_hwq8_split.txt
