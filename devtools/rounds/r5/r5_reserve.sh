#!/bin/bash
# what MI355XQR_TSQR_RESERVE_CUS costs the local factorisation: kernel averages with and without the mask
R=gpurun_out/r5_reserve; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 32; do
  export MI355XQR_TSQR_RESERVE_CUS=$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/s$c -o tl -- python3 devtools/tools_tsqr_latency.py 262144x512x8x128 > $R/log$c.txt 2>&1
  echo "RESERVE_CUS=$c"; grep -v amdgpu.ids $R/log$c.txt | cut -c1-200
  python3 - <<PY
import csv, glob
f = glob.glob("$R/s$c/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us avg', ('%.1f' % (float(r['TotalDurationNs']) / 1e6)).rjust(8), 'ms')
PY
  rm -rf $R/s$c
done
