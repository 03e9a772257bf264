#!/bin/bash
# round-4 evidence run: bench lines, rocprofv3 kernel stats of the same bench commands, config sweep, schedule Gantt, late-phase kernel
# timeline, one-launch panel: kernel alone + A/B inside whole factorisations, full-width tall panel: kernel times + A/B, comparator,
# host-pointer timing table with the vendor line.  PMC passes: devtools/rounds/r4/scripts_r4_pmc.sh.  Usage: scripts_r4_evidence.sh <git head>
HEAD=$1
R=gpurun_out/prof_r04; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
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
grep "gemm_nt_kernel\|gemm_tn_kernel<4, 4, true, 1>\|w8\|panel_fused" $R/c3/bench_kernel_stats.csv | cut -c1-160
python3 devtools/tools_perf.py 4096x4096x64 4096x4096x128 8192x8192x256 16384x16384x32 16384x16384x64 16384x16384x128 16384x16384x256 16384x16384x512 131072x256x128 65536x256x128 262144x256x128 262144x512x128 2097152x512x128 4096x512x128 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(json.dumps({'m': d['m'], 'n': d['n'], 'nb': d['nb'], 'ms': round(d['ms'], 2), 'gflops': round(d['tflops'] * 1e3, 1), 'panel_ms': round(d.get('panel', {}).get('ms', 0), 2)}))
" > $R/baseline_config_sweep.txt; cat $R/baseline_config_sweep.txt
date +%T
# schedule of C3 from the plan's own event records, and the kernel timeline of its chain-bound phase
python3 devtools/tools_gantt.py 16384x16384x256 2>/dev/null > $R/c3_gantt.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o tl -- python3 devtools/tools_one.py 16384x16384x256 > $R/tl_log.txt 2>&1
f=$(find $R/t -name "*kernel_trace.csv" | head -1)
( echo "# C3 16384^2 nb 256 under rocprofv3 --kernel-trace: queue 2 = panel stream (32 CUs), queue 3 = update stream (224 CUs)."
  echo "# (a) window at 93.5 % of the run: one outer step of the chain-bound phase with the one-launch panel (panel_fused_kernel, grid = (rows/256 + 1) x 256);"
  echo "# under the profiler the fused launch reads 0.53 ms where the plan's own events give 0.45-0.5 (r04_c3_schedule_gantt.txt)"
  python3 devtools/tools_trace_timeline.py $f 0.935 0.9
  echo; echo "# (b) window at 98.5 %: below MI355XQR_FUSED_MIN_ROWS = 3072 rows the launch chain (6 launches per leaf) is as fast and stays"
  python3 devtools/tools_trace_timeline.py $f 0.985 0.5 ) > $R/c3_panel_stream_timeline.txt
rm -f $f
date +%T
# the one-launch panel: kernel alone, and inside whole factorisations against the launch chain
PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 2>&1 | grep -v amdgpu.ids > $R/panel_fused_perf.txt
bash devtools/rounds/r4/r4_fused_ab.sh > $R/fused_ab.txt 2>&1
bash devtools/rounds/r4/r4_split.sh > $R/cu_split_tail.txt 2>&1
date +%T
# the full-width tall panel: kernel times of one 262144 x 128 panel, and against the leaf chain
bash devtools/rounds/r4/r4_cqr_e2.sh 262144 128 > $R/cqr_kernel_times.txt 2>&1
bash devtools/rounds/r4/r4_cqr.sh > $R/cqr_vs_leaves.txt 2>&1
date +%T
( echo "# qr_device timing table (this build, fp64, MI355X) at the nominal sizes of the reference's timing.txt; --compare adds the rocSOLVER line (qr.cu:790-806)";
for mm in 256 512 1024 2048 4096 8192 16384 32768 65536 131072; do ./cuda-qr_amd/build/qr_device $mm 64 | grep "MMQR ran"; done
for mm in 64 128 256 512 1024 2048 4096; do ./cuda-qr_amd/build/qr_device $mm $mm | grep "MMQR ran"; done
for s in "8192 8192" "16384 16384" "262144 512"; do ./cuda-qr_amd/build/qr_device $s --compare | grep "ran QR"; done ) 2>&1 | grep -v amdgpu.ids > $R/qr_device_timing_table.txt
tail -9 $R/qr_device_timing_table.txt
python3 devtools/tools_applyq.py 2>&1 | grep -v amdgpu.ids > $R/form_q_timing.txt
python3 devtools/tools_comparator.py 2>&1 | grep -v amdgpu.ids > $R/comparator_rocsolver.txt; tail -5 $R/comparator_rocsolver.txt
python3 devtools/tools_fuzz_parity.py > $R/fuzz_parity.txt 2>&1; tail -3 $R/fuzz_parity.txt
date +%T
