#!/bin/bash
# round 6, run 21: the new block-size / look-ahead rule (lookahead_pays, default_blocks): what the library now picks by itself (MxNx0), next to
# the three block sizes forced; full GPU suite first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run21; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
S=""
for s in 512x512 1024x1024 1536x1536 2048x2048 2560x2560 2816x2816 3072x3072 3584x3584 4096x4096 6144x6144 8192x8192 2048x1024 3072x1024 4096x1024 8192x1024 3072x2048 4096x2048 5120x2048 6144x2048 8192x2048 10240x2048 12288x2048 16384x2048 32768x2048 4096x3072 6144x3072 12288x3072 8192x4096 12288x4096 16384x4096 20480x4096 24576x4096 32768x4096 16384x8192 32768x8192; do S="$S ${s}x0"; done
( for i in 1 2; do python3 devtools/tools_perf.py $S 2>&1 | fmt; done ) > $O/default_rule.txt 2>&1
cat $O/default_rule.txt
