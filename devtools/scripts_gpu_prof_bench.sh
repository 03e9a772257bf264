#!/bin/bash
R=gpurun_out/prof_r01
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf $R/bench_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_c3 -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err
python3 devtools/tools_trace_summary.py $R/bench_c3/bench_kernel_trace.csv > $R/bench_c3_trace_summary.txt; rm -f $R/bench_c3/bench_kernel_trace.csv
python -c "
import json; d=json.load(open('$R/bench_c3.json')); print('C3', d['value'], d['ms_per_step'], d['accuracy'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'], d['roofline']['companion_tn']['achieved'], d['weak_scaling_base_1gpu'])"
grep "w8\|tn_kernel<4, 4, true, 1>" $R/bench_c3/bench_kernel_stats.csv | cut -c1-200
