#!/bin/bash
# round 6, run 4: why does a second CU-partition phase (more CUs for the panel chain in the tail) cost ~0.28 ms per step (r04_cu_split_tail.txt)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run4; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
export CUDA_QR_AMD_LIB=lab
( echo "== default"; python3 devtools/tools_perf.py 16384x16384x256 2>&1 | fmt
  for sp in "32:0.25,64" "32:0.2,64" "32:0.2,48"; do
    echo "== MI355XQR_SPLIT=$sp"; MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py 16384x16384x256 2>&1 | fmt
    echo "== MI355XQR_SPLIT=$sp MI355XQR_BALANCE=7.04,51.5,1.1,0.6"; MI355XQR_BALANCE=7.04,51.5,1.1,0.6 MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py 16384x16384x256 2>&1 | fmt
  done ) > $O/split_ab.txt 2>&1
cat $O/split_ab.txt
MI355XQR_SPLIT=32:0.25,64 python3 devtools/tools_gantt.py 16384x16384x256 > $O/gantt_split64.txt 2>&1
MI355XQR_SPLIT=32:0.25,64 MI355XQR_BALANCE=7.04,51.5,1.1,0.6 python3 devtools/tools_gantt.py 16384x16384x256 > $O/gantt_split64_bal.txt 2>&1
tail -24 $O/gantt_split64.txt
