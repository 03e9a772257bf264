#!/bin/bash
# round 6, run 24: full GPU suite on the new block-size rule; then MI355XQR_SMALL_T (lab): small wide updates through the T-folded reduction
# (three launches, no V*T) -- threshold in Ki elements of the trailing block
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run24; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), 'resid', d.get('resid'))
"; }
S="512x512x64 1024x1024x64 1536x1536x64 2048x2048x64 2560x2560x64 4096x4096x64 2048x1024x64 1024x1024x128 2048x2048x128 4096x4096x128"
( for t in 0 256 1024 4096 16384 0 256 1024 4096 16384; do
  echo "== MI355XQR_SMALL_T=$t"; CHECK=1 CUDA_QR_AMD_LIB=lab MI355XQR_LOOKAHEAD=0 MI355XQR_SMALL_T=$t python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/small_t.txt 2>&1
python3 - <<'PY'
import collections
best = collections.defaultdict(dict); sp = None
for l in open("gpurun_out/r6_run24/small_t.txt"):
    if l.startswith("=="): sp = l.strip().split("=")[-1]; continue
    t = l.split()
    try: m, n, nb, ms = int(t[0]), int(t[1]), int(t[2]), float(t[4])
    except Exception: print(l.strip()); continue
    best[(m, n, nb)][sp] = min(best[(m, n, nb)].get(sp, 1e9), ms)
for k, d in best.items():
    print(*k, "  ".join("%s %.3f" % (s, v) for s, v in d.items()), " best", min(d, key=d.get))
PY
grep resid $O/small_t.txt | awk '{print $NF}' | sort | uniq -c | sort -rn | head -5
