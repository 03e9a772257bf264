#!/bin/bash
# round 6, run 48: what a width that is not a multiple of 32 costs (the last panel is ragged: leaf chain instead of one launch)
cd $GRAFT_REPO_ROOT
python3 devtools/tools_perf.py 8192x8192x0 8192x8191x0 8192x8160x0 8192x8100x0 5000x3008x0 5000x3000x0 5000x2976x0 4096x4000x0 4096x4001x0 4096x4032x0 2048x2000x0 2048x2016x0 65536x500x0 65536x512x0 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"
