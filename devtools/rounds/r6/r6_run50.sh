#!/bin/bash
# round 6, run 50: 262144 x 500 against 262144 x 512 by class and by kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run50; mkdir -p $O
python3 devtools/tools_perf.py 262144x500x0 262144x512x0 262144x448x0 262144x480x0 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[-300:]); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], {k: (v['ms'], v['n']) for k, v in d.items() if isinstance(v, dict)})
"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t500 -o t -- python3 devtools/tools_perf.py 262144x500x0 > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r6_run50/t500/t_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]: print("%-80s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
rm -f $O/t500/t_kernel_trace.csv
