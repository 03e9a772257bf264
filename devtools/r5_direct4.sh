#!/bin/bash
R=gpurun_out/r5_direct4; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mk in 4096 70003 256 48 8200; do MI355XQR_CQR_DIRECT=1 timeout -k 10 120 python3 devtools/tools_cqr_direct_debug.py $mk 2>&1 | grep -v amdgpu.ids; done
export MI355XQR_CQR_DIRECT=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o tl -- python3 devtools/tools_cqr_perf.py 262144 128 0 > $R/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/s/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    if 'cqr' in r['Name']: print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us avg')
PY
rm -rf $R/s
