#!/bin/bash
# the small-factor core after a change: parity, its own timing, the one-launch panel, the one-workgroup kernels of the full-width panel, shapes
R=gpurun_out/r5_lu32; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_factor32.py tests/test_gpu_panel_cqr.py -x -q > $R/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $R/tests.txt
timeout -k 10 120 python3 devtools/tools_factor32.py 2>&1 | grep -v amdgpu.ids | grep "1 wave"
PF_NO_GRAM=1 timeout -k 10 200 python3 devtools/tools_panel_fused_perf.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o tl -- python3 devtools/tools_cqr_perf.py 262144 128 0 > $R/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/s/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if 'cqr_lu' in r['Name'] or 'cqr_chol' in r['Name']: print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us avg')
PY
rm -rf $R/s
timeout -k 10 300 python3 devtools/tools_perf.py 4096x4096x64 4096x4096x128 8192x8192x256 65536x256x128 262144x512x128 16384x16384x256 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'])
"
