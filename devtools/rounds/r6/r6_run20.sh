#!/bin/bash
# round 6, run 20: block size / look-ahead rule, the rest of the boundary (tiny squares, 5:1 ... 8:1 shapes with many columns)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run20; mkdir -p $O
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S=""
for s in 256x256 512x512 768x768 1024x512 2048x512 2048x1024 3072x1024 3072x2048 2816x2816 10240x2048 20480x4096 24576x4096 32768x8192 49152x8192 65536x8192 12288x3072 5120x2048 3072x1536 2048x1536; do for nb in 64 128 256; do S="$S ${s}x$nb"; done; done
( for la in 0 1 0 1; do
  echo "== MI355XQR_LOOKAHEAD=$la"; MI355XQR_LOOKAHEAD=$la python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/nb_lookahead_rule2.txt 2>&1
python3 - <<'PY'
import collections
best = collections.defaultdict(dict); la = None
for l in open("gpurun_out/r6_run20/nb_lookahead_rule2.txt"):
    if l.startswith("=="): la = int(l.strip()[-1]); continue
    t = l.split()
    try: m, n, nb, ms = int(t[0]), int(t[1]), int(t[2]), float(t[4])
    except Exception: continue
    k = (nb, la); best[(m, n)][k] = min(best[(m, n)].get(k, 1e9), ms)
for (m, n), d in best.items():
    print(m, n, "  ".join("nb%d/la%d %.2f" % (k[0], k[1], v) for k, v in sorted(d.items())), " best", min(d, key=d.get))
PY
