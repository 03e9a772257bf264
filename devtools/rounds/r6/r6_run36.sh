#!/bin/bash
# round 6, run 36: off-grid sizes next to C3 (16448 = 64 mod 128, 16400 = 16 mod 128, 16512 = 128 mod 256): where does the time go, by class
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run36; mkdir -p $O
python3 devtools/tools_perf.py 16384x16384x256 16448x16448x256 16512x16512x256 16400x16400x256 16384x16400x256 16400x16384x256 8200x8200x256 4100x4100x256 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], {k: (v['ms'], v['tflops']) for k, v in d.items() if isinstance(v, dict)})
" > $O/offgrid.txt 2>&1
cat $O/offgrid.txt
