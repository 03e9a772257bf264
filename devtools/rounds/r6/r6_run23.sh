#!/bin/bash
# round 6, run 23: (i) MI355XQR_SPLIT=48 no longer times the one-launch panel out; (ii) block size on tall and on tiny shapes (32 / 64 / 96 / 128 / 256)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run23; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_qr.py tests/test_gpu_panel_fused.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S=""
for s in 16384x256 32768x256 65536x256 16384x512 32768x512 65536x512 131072x512 8192x512 4096x512 16384x1024 32768x1024 65536x1024 131072x1024 65536x2048; do for nb in 64 128 256; do S="$S ${s}x$nb"; done; done
for s in 256x256 512x512 768x768 1024x1024 1536x1536 2048x2048 1024x512 2048x1024; do for nb in 32 64 96 128; do S="$S ${s}x$nb"; done; done
( for i in 1 2; do python3 devtools/tools_perf.py $S 2>&1 | fmt; done ) > $O/nb_tall_tiny.txt 2>&1
python3 - <<'PY'
import collections
best = collections.defaultdict(dict)
for l in open("gpurun_out/r6_run23/nb_tall_tiny.txt"):
    t = l.split()
    try: m, n, nb, ms = int(t[0]), int(t[1]), int(t[2]), float(t[4])
    except Exception: print(l.strip()); continue
    best[(m, n)][nb] = min(best[(m, n)].get(nb, 1e9), ms)
for (m, n), d in best.items():
    print(m, n, "  ".join("nb%d %.3f" % (k, v) for k, v in sorted(d.items())), " best", min(d, key=d.get))
PY
