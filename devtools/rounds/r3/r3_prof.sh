#!/bin/bash
# per-kernel stats of one configuration: r3_prof.sh <tag> <cfg...>   (env passes through)
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/devtools/tools_perf.py "$@" > $out/run.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-58s n=%5s avg=%9.1f us tot=%8.2f ms" % (r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
