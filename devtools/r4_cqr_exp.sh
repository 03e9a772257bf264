#!/bin/bash
# which side bounds the streaming passes: experimental builds without the Gram instructions (1) / without the global loads (2)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for e in 1 2; do
  R=gpurun_out/cqr_exp$e; rm -rf $R; mkdir -p $R
  CQR_LIB=cuda-qr_amd/libmi355xqr_exp$e.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o tl -- python3 devtools/tools_cqr_perf.py 262144 128 0 > $R/log.txt 2>&1
  echo "== CS_EXP=$e"
  python3 - <<PY
import csv, glob
f = glob.glob("$R/s/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    if "cqr_" in r['Name']: print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us avg')
PY
done
