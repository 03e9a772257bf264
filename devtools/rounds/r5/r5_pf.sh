#!/bin/bash
# one-launch panel after moving the deferred update behind the G2 publish: parity tests, kernel alone, phase stamps, whole shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_panel_fused.py tests/test_gpu_multipanel_golden.py -x -q > gpurun_out/r5_pf_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5_pf_tests.txt
PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 2>&1 | grep -v amdgpu.ids | tail -9
python3 devtools/tools_panel_fused_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_pf_stamps2.txt; grep "leaf 0 start" gpurun_out/r5_pf_stamps2.txt
python3 devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 4096x4096x128 4096x512x128 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], {k: round(v['ms'],2) for k, v in d.items() if isinstance(v, dict) and 'ms' in v})
"
