#!/bin/bash
# round 6, run 19: default block size and look-ahead rule over small squares and 2:1 ... 16:1 shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run19; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S=""
for s in 1024x1024 1536x1536 2048x2048 2560x2560 3072x3072 3584x3584 4096x2048 6144x2048 8192x2048 12288x2048 16384x2048 32768x2048 8192x4096 12288x4096 16384x4096 32768x4096 6144x3072 4096x3072 8192x1024 4096x1024 16384x8192; do for nb in 64 128 256; do S="$S ${s}x$nb"; done; done
( for la in 0 1 0 1; do
  echo "== MI355XQR_LOOKAHEAD=$la"; MI355XQR_LOOKAHEAD=$la python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/nb_lookahead_rule.txt 2>&1
python3 - <<'PY'
import collections
best = collections.defaultdict(dict); la = None
for l in open("gpurun_out/r6_run19/nb_lookahead_rule.txt"):
    if l.startswith("=="): la = int(l.strip()[-1]); continue
    t = l.split()
    try: m, n, nb, ms = int(t[0]), int(t[1]), int(t[2]), float(t[4])
    except Exception: continue
    k = (nb, la); best[(m, n)][k] = min(best[(m, n)].get(k, 1e9), ms)
for (m, n), d in best.items():
    print(m, n, "  ".join("nb%d/la%d %.2f" % (k[0], k[1], v) for k, v in sorted(d.items())), " best", min(d, key=d.get))
PY
