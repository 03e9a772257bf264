#!/bin/bash
# round 6, run 49: a ragged last panel split into its whole leaves + the rest (single-stream schedule): tests, then tall and small shapes with ragged widths
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run49; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
( timeout -k 5 300 python3 devtools/tools_fuzz_parity.py 3 ragged; timeout -k 5 300 python3 devtools/tools_fuzz_parity.py 6 edges6; timeout -k 5 300 python3 devtools/tools_cqr_fuzz.py ) 2>&1 | grep -v amdgpu.ids > $O/fuzz.txt; grep -c " x " $O/fuzz.txt; grep "^ok\|Error\|assert" $O/fuzz.txt
python3 devtools/tools_perf.py 65536x500x0 65536x512x0 262144x500x0 262144x512x0 100000x300x0 100000x320x0 2048x2000x0 2048x2016x0 1000x1000x0 1024x1024x0 10000x1000x0 10000x1024x0 8192x8191x0 8192x8192x0 5000x3000x0 5000x3008x0 33001x97x0 70001x321x0 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
" | tee $O/ragged_width.txt
