#!/bin/bash
# round 6, run 22: CU partition of the look-ahead schedule (MI355XQR_SPLIT) on the shapes that newly run two-stream, and on the old ones
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run22; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S=""
for s in 3072x3072 4096x4096 6144x6144 8192x8192 10240x10240 12288x12288 4096x2048 6144x2048 8192x2048 4096x3072 8192x4096 12288x4096 16384x4096 16384x8192 32768x8192; do S="$S ${s}x0"; done
( for sp in default 32 48 64 96 default 32 48 64 96; do
  echo "== MI355XQR_SPLIT=$sp"
  if [ $sp = default ]; then python3 devtools/tools_perf.py $S 2>&1 | fmt; else MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py $S 2>&1 | fmt; fi
  done ) > $O/split.txt 2>&1
python3 - <<'PY'
import collections
best = collections.defaultdict(dict); sp = None
for l in open("gpurun_out/r6_run22/split.txt"):
    if l.startswith("=="): sp = l.strip().split("=")[-1]; continue
    t = l.split()
    try: m, n, nb, ms = int(t[0]), int(t[1]), int(t[2]), float(t[4])
    except Exception: continue
    best[(m, n, nb)][sp] = min(best[(m, n, nb)].get(sp, 1e9), ms)
for k, d in best.items():
    print(*k, "  ".join("%s %.2f" % (s, v) for s, v in d.items()), " best", min(d, key=d.get))
PY
