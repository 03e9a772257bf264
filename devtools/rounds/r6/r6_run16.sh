#!/bin/bash
# round 6, run 16: where the update still outlasts the panel (C3 steps 30-47) the look-ahead update sits between two wide updates on the
# update stream (a 0.11 ms bubble per step); on the panel stream's idle CUs instead?  lab knobs MI355XQR_NEXT=auto + MI355XQR_TAILTC (flat chain model)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run16; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
export CUDA_QR_AMD_LIB=lab
S="16384x16384x256 16384x16384x256 12288x12288x256"
( echo "== default"; python3 devtools/tools_perf.py $S 2>&1 | fmt
  for tc in 0 0.6 0.7 0.8 0.9 1.0; do echo "== MI355XQR_NEXT=auto MI355XQR_TAILTC=$tc"; MI355XQR_NEXT=auto MI355XQR_TAILTC=$tc python3 devtools/tools_perf.py $S 2>&1 | fmt; done
  echo "== MI355XQR_NEXT=panel"; MI355XQR_NEXT=panel python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== default"; python3 devtools/tools_perf.py $S 2>&1 | fmt ) > $O/next_ab.txt 2>&1
cat $O/next_ab.txt
MI355XQR_NEXT=auto MI355XQR_TAILTC=0.8 python3 devtools/tools_gantt.py 16384x16384x256 2>/dev/null > $O/gantt_auto_08.txt
sed -n 30,55p $O/gantt_auto_08.txt
