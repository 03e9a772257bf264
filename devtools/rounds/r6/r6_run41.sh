#!/bin/bash
# round 6, run 41: an UNMASKED panel stream for the tail ("U" phase of MI355XQR_SPLIT): with 128-row workgroups the one-launch panel can use more
# than the 32 compute units of the panel partition once the update stream idles half of the time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run41; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S="16384x16384x256 8192x8192x256 12288x12288x256 10240x10240x256"
( for sp in default "32:0.3,U" "32:0.5,U" "32:0.7,U" default "32:0.3,U" "32:0.5,U" "32:0.7,U"; do
  echo "== MI355XQR_SPLIT=$sp"
  if [ "$sp" = default ]; then python3 devtools/tools_perf.py $S 2>&1 | fmt; else MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py $S 2>&1 | fmt; fi
  done ) > $O/split_u.txt 2>&1
cat $O/split_u.txt
MI355XQR_SPLIT=32:0.5,U python3 devtools/tools_gantt.py 8192x8192x256 2>/dev/null > $O/gantt_8192_u.txt; tail -24 $O/gantt_8192_u.txt
