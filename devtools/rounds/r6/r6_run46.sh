#!/bin/bash
# round 6, run 46: plans of heights that are not multiples of 16 factor a zero-padded copy (device API): full GPU suite, fuzz, odd sizes again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run46; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
( timeout -k 5 300 python3 devtools/tools_fuzz_parity.py 3 ragged; timeout -k 5 300 python3 devtools/tools_fuzz_parity.py 6 edges6 ) 2>&1 | grep -v amdgpu.ids > $O/fuzz.txt; grep -c " x " $O/fuzz.txt; grep "^ok\|Error\|assert" $O/fuzz.txt
python3 devtools/tools_perf.py 5000x5000x0 5001x5001x0 5002x5002x0 8191x8191x0 8192x8192x0 10001x3001x0 10000x3000x0 777x555x0 100001x500x0 100000x500x0 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'])
" | tee $O/odd_device_api.txt
