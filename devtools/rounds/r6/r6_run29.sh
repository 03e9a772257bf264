#!/bin/bash
# round 6, run 29: with one-launch panels up to 16384 rows (MI355XQR_PF_MAX_ROWS=16384, lab): block size, look-ahead, full-width threshold and CU split again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run29; mkdir -p $O
export CUDA_QR_AMD_LIB=lab MI355XQR_PF_MAX_ROWS=16384
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S=""
for s in 10240x512 12288x512 16384x512 10240x1024 12288x1024 16384x1024 10240x2048 12288x2048 16384x2048 10240x3072 12288x3072 16384x3072 10240x4096 12288x4096 16384x4096 20480x4096 16384x6144; do for nb in 128 256; do S="$S ${s}x$nb"; done; done
( for cq in 8193 16385; do for la in 0 1; do
  echo "== MI355XQR_CQR_MIN_ROWS=$cq MI355XQR_LOOKAHEAD=$la"; MI355XQR_CQR_MIN_ROWS=$cq MI355XQR_LOOKAHEAD=$la python3 devtools/tools_perf.py $S 2>&1 | fmt
  done; done ) > $O/tall_rules.txt 2>&1
python3 - <<'PY'
import collections
best = collections.defaultdict(dict); key = None
for l in open("gpurun_out/r6_run29/tall_rules.txt"):
    if l.startswith("=="):
        t = l.split(); key = (int(t[1].split("=")[1]), int(t[2].split("=")[1])); continue
    t = l.split()
    try: m, n, nb, ms = int(t[0]), int(t[1]), int(t[2]), float(t[4])
    except Exception: print(l.strip()); continue
    best[(m, n)][(nb,) + key] = ms
for (m, n), d in best.items():
    print(m, n, "  ".join("nb%d/cq%d/la%d %.2f" % (k[0], k[1] // 1000, k[2], v) for k, v in sorted(d.items())), " best", min(d, key=d.get))
PY
S2="10240x10240x256 12288x12288x256 16384x16384x256 16384x8192x256 14336x14336x256 16384x12288x256"
( for sp in 32 64 32 64; do echo "== MI355XQR_SPLIT=$sp"; MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py $S2 2>&1 | fmt; done ) > $O/split_again.txt 2>&1
cat $O/split_again.txt
