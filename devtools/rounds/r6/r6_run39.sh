#!/bin/bash
# round 6, run 39: the same shape several times in one process (a new plan each time): do later plans run slower (hardware queues of destroyed streams)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run39; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'])
"; }
( echo "== 6 x 6000^2"; python3 devtools/tools_perf.py 6000x6000x0 6000x6000x0 6000x6000x0 6000x6000x0 6000x6000x0 6000x6000x0 2>&1 | fmt
  echo "== 2 single-stream plans, then 6000^2 x 3"; python3 devtools/tools_perf.py 2048x2048x0 4096x4096x64 6000x6000x0 6000x6000x0 6000x6000x0 2>&1 | fmt
  echo "== the same with GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 python3 devtools/tools_perf.py 2048x2048x0 4096x4096x64 6000x6000x0 6000x6000x0 6000x6000x0 2>&1 | fmt
  echo "== the same with GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 python3 devtools/tools_perf.py 2048x2048x0 4096x4096x64 6000x6000x0 6000x6000x0 6000x6000x0 2>&1 | fmt
  echo "== 16384^2 after five other plans"; python3 devtools/tools_perf.py 2048x2048x0 4096x4096x64 6000x6000x0 8192x8192x0 3072x3072x0 16384x16384x256 16384x16384x256 2>&1 | fmt
) > $O/plan_order.txt 2>&1
cat $O/plan_order.txt
