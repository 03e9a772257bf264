#!/bin/bash
# round-2 evidence run: full GPU test suite, bench lines, rocprofv3 kernel stats of the same bench commands, config sweep,
# host-pointer timing table.  PMC passes: devtools/rounds/r2/scripts_r2_pmc.sh.  Usage: scripts_r2_evidence.sh <git head>
HEAD=$1
R=gpurun_out/prof_r02; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
date +%T
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -4 $R/tests.log
date +%T
python3 bench.py --steps 10 --warmup 3 > $R/bench_c3.json 2> $R/bench_c3.err; echo "bench c3 rc=$?"
python3 bench.py --workload tsqr --steps 10 --warmup 3 > $R/bench_tsqr.json 2> $R/bench_tsqr.err; echo "bench tsqr rc=$?"
python3 bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_c2.json 2> $R/bench_c2.err; echo "bench c2 rc=$?"
date +%T
rocprofv3 --kernel-trace --stats --output-format csv -d $R/c3 -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/bench_c3_under_rocprof.json 2> $R/c3.err
python3 devtools/tools_trace_summary.py $R/c3/bench_kernel_trace.csv > $R/bench_c3_trace_summary.txt; rm -f $R/c3/bench_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/tsqr -o bench -- python3 bench.py --workload tsqr --steps 3 --warmup 1 --no-cpu-baseline > $R/bench_tsqr_under_rocprof.json 2> $R/tsqr.err
rm -f $R/tsqr/bench_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/c2 -o bench -- python3 bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $R/bench_c2_under_rocprof.json 2> $R/c2.err
rm -f $R/c2/bench_kernel_trace.csv
date +%T
python3 - <<PY
import json
for n in ("bench_c3", "bench_tsqr", "bench_c2", "bench_c3_under_rocprof"):
    try:
        d = json.loads(open("$R/%s.json" % n).read().strip().splitlines()[-1]); r = d["roofline"]
        print(n, "value %.0f GF/s  ms %.2f  acc %s  roof %s %.3g %s frac %.3f" % (d["value"], d["ms_per_step"], d["accuracy"], r["bound"], r["achieved"], r["unit"], r["frac"]))
    except Exception as e:
        print(n, "parse failed", e)
PY
grep "gemm_nt_kernel\|gemm_tn_kernel<4, 4, true, 1>\|w8" $R/c3/bench_kernel_stats.csv | cut -c1-160
# BASELINE configs: C2 (4096^2, nb 64), C3 block-size sweep, one C4 shard each for P = 2 and 4, C4 whole, one C5 shard, C5 whole
python3 devtools/tools_perf.py 4096x4096x64 4096x4096x128 8192x8192x256 16384x16384x32 16384x16384x64 16384x16384x128 16384x16384x256 16384x16384x512 131072x256x128 65536x256x128 262144x256x128 262144x512x128 2097152x512x128 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(json.dumps({'m': d['m'], 'n': d['n'], 'nb': d['nb'], 'ms': round(d['ms'], 2), 'gflops': round(d['tflops'] * 1e3, 1), 'panel_ms': round(d.get('panel', {}).get('ms', 0), 2)}))
" > $R/baseline_config_sweep.txt; cat $R/baseline_config_sweep.txt
( echo "# qr_device timing table (this build, fp64, MI355X) at the nominal sizes of the reference's timing.txt";
for mm in 256 512 1024 2048 4096 8192 16384 32768 65536 131072; do ./cuda-qr_amd/build/qr_device $mm 64 | grep "MMQR ran"; done
for mm in 64 128 256 512 1024 2048 4096; do ./cuda-qr_amd/build/qr_device $mm $mm | grep "MMQR ran"; done ) 2>&1 | grep -v amdgpu.ids > $R/qr_device_timing_table.txt
head -8 $R/qr_device_timing_table.txt
python3 devtools/tools_gantt.py 16384x16384x256 2>/dev/null > $R/c3_gantt.txt
python3 devtools/tools_applyq.py 2>&1 | grep -v amdgpu.ids > $R/form_q_timing.txt
python3 devtools/tools_comparator.py 2>&1 | grep -v amdgpu.ids > $R/comparator_rocsolver.txt; tail -5 $R/comparator_rocsolver.txt
date +%T
