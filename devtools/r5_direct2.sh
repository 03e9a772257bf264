#!/bin/bash
R=gpurun_out/r5_direct2; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 2 3; do echo DIRECT=$d; MI355XQR_CQR_DIRECT=$d timeout -k 10 120 python3 devtools/tools_cqr_direct_debug.py 4096 2>&1 | grep -v amdgpu.ids; done
MI355XQR_CQR_DIRECT=3 timeout -k 10 120 python3 devtools/tools_cqr_direct_debug.py 70003 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python3 -m pytest tests/test_gpu_panel_cqr.py -x -q > $R/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $R/tests.txt
for x in 1 2; do
  export MI355XQR_CD_EXP=$x
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/s$x -o tl -- python3 devtools/tools_cqr_perf.py 262144 128 0 > $R/log$x.txt 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$R/s$x/*kernel_stats.csv")[0]
print("MI355XQR_CD_EXP=$x")
for r in list(csv.DictReader(open(f)))[:12]:
    if 'cqr' in r['Name']: print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us avg')
PY
  rm -rf $R/s$x
done
