#!/bin/bash
# round 6, run 38: half tiles not at K = 32 (C3 at nb 32 went 296 -> 324 ms through them); re-check nb 32 / 64 at C3 and C2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run38; mkdir -p $O
python3 devtools/tools_perf.py 16384x16384x32 16384x16384x64 4096x4096x32 4096x4096x64 8192x8192x32 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'])
" > $O/nb32.txt; cat $O/nb32.txt
