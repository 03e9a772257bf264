#!/bin/bash
# kernel times of one full-width tall panel under rocprofv3 (devtools/tools_cqr_perf.py)
R=gpurun_out/cqr_e2; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o tl -- python3 devtools/tools_cqr_perf.py ${1:-262144} ${2:-128} 0 > $R/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/s/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us avg')
PY
