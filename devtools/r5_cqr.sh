#!/bin/bash
# round 5: the full-width tall panel after the matrix-core diagonal blocks: tests, kernel times under rocprofv3, crossover
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_panel_cqr.py tests/test_gpu_factor32.py -x -q -m gpu > gpurun_out/cqr_test.txt 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/cqr_test.txt
bash devtools/r4_cqr_e2.sh 262144 128 > gpurun_out/r5_cqr_kernel_times.txt 2>&1
cat gpurun_out/r5_cqr_kernel_times.txt
S="262144x512x128 131072x256x128 65536x256x128"
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
echo "== default"; python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== MI355XQR_CQR_MIN_ROWS=32768"; MI355XQR_CQR_MIN_ROWS=32768 python3 devtools/tools_perf.py $S 2>&1 | fmt
echo "== MI355XQR_GUARD=latch"; MI355XQR_GUARD=latch python3 devtools/tools_perf.py $S 2>&1 | fmt
