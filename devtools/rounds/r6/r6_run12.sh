#!/bin/bash
# round 6, run 12: in-launch T merge of 64-column panels: parity + C2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run12; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_panel_fused.py tests/test_gpu_multipanel_golden.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
CHECK=1 python3 devtools/tools_perf.py 4096x4096x64 4096x4096x64 2048x2048x64 1024x1024x64 8192x8192x64 4096x512x64 2>&1 | fmt > $O/c2.txt; cat $O/c2.txt
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; python3 -c "
import json; d = json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('bench c2 ms %.3f' % d['ms_per_step'], d['accuracy'])"
