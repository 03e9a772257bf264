#!/bin/bash
# per-kernel times of one tall-skinny shard factorisation under rocprofv3
R=gpurun_out/prof_shard; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o tl -- python3 devtools/tools_perf.py ${1:-262144x512x128} > $R/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/s/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("calls are over 3 factorisations (+ fills); total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:22]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(9), 'us avg', ('%.2f' % (float(r['TotalDurationNs']) / 1e6)).rjust(8), 'ms total', ('%.1f%%' % (100 * float(r['TotalDurationNs']) / tot)).rjust(6))
PY
rm -f $R/s/*kernel_trace.csv
