#!/bin/bash
# the full-width tall panel: parity tests, kernel times of one 262144 x 128 panel under rocprofv3, whole shapes (the lab knob
# MI355XQR_CQR_DIRECT this script once swept lived in commit b04e155 only: profiles/r05_cqr_direct_passes.txt)
R=gpurun_out/r5_direct; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_panel_cqr.py -x -q > $R/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $R/tests.txt
for d in 0; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/s$d -o tl -- python3 devtools/tools_cqr_perf.py 262144 128 0 > $R/log$d.txt 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$R/s$d/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if 'cqr' in r['Name']: print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us avg')
PY
  rm -rf $R/s$d
  timeout -k 10 300 python3 devtools/tools_perf.py 262144x512x128 131072x256x128 65536x256x128 16384x16384x256 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(json.dumps({'m': d['m'], 'n': d['n'], 'nb': d['nb'], 'ms': round(d['ms'], 3), 'resid': d.get('resid'), 'orth': d.get('orth')}))
"
done
