#!/bin/bash
# round-6 evidence run: bench lines (incl. the ill-conditioned input: the price of a refused panel), rocprofv3 kernel stats of the same bench
# commands, config sweep, schedule Gantt, one-launch panel (perf table for both row splits, phase stamps), full-width tall panel kernel times,
# rank-step latencies, guard price, fuzz parity, timing table.  PMC passes: devtools/rounds/r6/scripts_r6_pmc.sh.  Usage: scripts_r6_evidence.sh <git head>
HEAD=$1
R=gpurun_out/prof_r06; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
date +%T
python3 bench.py --steps 20 --warmup 5 > $R/bench_c3.json 2> $R/bench_c3.err; echo "bench c3 rc=$?"
python3 bench.py --workload tsqr --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_tsqr.json 2> $R/bench_tsqr.err; echo "bench tsqr rc=$?"
python3 bench.py --workload tsqr --cond 1e9 --steps 5 --warmup 2 --no-cpu-baseline > $R/bench_tsqr_cond1e9.json 2> $R/bench_tsqr_cond.err; echo "bench tsqr cond rc=$?"
python3 bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $R/bench_c2.json 2> $R/bench_c2.err; echo "bench c2 rc=$?"
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
for n in ("bench_c3", "bench_tsqr", "bench_tsqr_cond1e9", "bench_c2", "bench_c3_under_rocprof"):
    try:
        d = json.loads(open("$R/%s.json" % n).read().strip().splitlines()[-1]); r = d["roofline"]
        print(n, "value %.0f GF/s  ms %.2f  acc %s  roof %s %.3g %s frac %.3f  routes %s" % (d["value"], d["ms_per_step"], d["accuracy"], r["bound"], r["achieved"], r["unit"], r["frac"], d.get("panel_routes")))
    except Exception as e:
        print(n, "parse failed", e)
PY
grep "gemm_nt4_kernel\|gemm_tn_kernel<4, 4, true, 1\|panel_fused\|trsm_gt" $R/c3/bench_kernel_stats.csv | cut -c1-170
python3 devtools/tools_perf.py 4096x4096x64 4096x4096x128 4096x4096x256 8192x8192x256 16384x16384x32 16384x16384x64 16384x16384x128 16384x16384x256 16384x16384x512 131072x256x128 65536x256x128 262144x256x128 262144x512x128 2097152x512x128 4096x512x128 2048x2048x128 1024x1024x128 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(json.dumps({'m': d['m'], 'n': d['n'], 'nb': d['nb'], 'ms': round(d['ms'], 2), 'gflops': round(d['tflops'] * 1e3, 1), 'panel_ms': round(d.get('panel', {}).get('ms', 0), 2)}))
" > $R/baseline_config_sweep.txt; cat $R/baseline_config_sweep.txt
date +%T
python3 devtools/tools_gantt.py 16384x16384x256 2>/dev/null > $R/c3_gantt.txt
( for r in 256 128 0; do PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py $r 2>&1 | grep -v amdgpu.ids; done ) > $R/panel_fused_perf.txt; tail -14 $R/panel_fused_perf.txt
python3 devtools/tools_panel_fused_stamps.py 2>&1 | grep -v amdgpu.ids > $R/panel_fused_stamps.txt; echo "stamps rc=$?"
bash devtools/rounds/r4/r4_cqr_e2.sh 262144 128 > $R/cqr_kernel_times.txt 2>&1; cat $R/cqr_kernel_times.txt
date +%T
SPECS="262144x512x8x128 65536x256x4x128 131072x256x2x128"
( python3 devtools/tools_tsqr_latency.py $SPECS; MI355XQR_TSQR_PIPE=0 python3 devtools/tools_tsqr_latency.py $SPECS ) 2>&1 | grep -v amdgpu.ids > $R/tsqr_rank_step_latency.txt; cat $R/tsqr_rank_step_latency.txt
( echo "# geqrf of tall shapes, drained per step: plain uniform input, then bench.py --cond's input (every 128-column panel = its first column + noise / cond): every";
  echo "# full-width panel refused and retried preconditioned (shifted CholeskyQR3); last: the same with the retry off (lab knob MI355XQR_CQR_RETRY=0 = round 5: leaf chain)";
  python3 devtools/tools_cond.py 0 262144x512x128 131072x256x128 65536x256x128; python3 devtools/tools_cond.py 1e9 262144x512x128 131072x256x128 65536x256x128;
  python3 devtools/tools_cond.py 1e5 262144x512x128 65536x256x128;
  CUDA_QR_AMD_LIB=lab MI355XQR_CQR_RETRY=0 python3 devtools/tools_cond.py 1e9 262144x512x128 131072x256x128 65536x256x128 | sed 's/^/retry off: /' ) 2>&1 | grep -v amdgpu.ids > $R/guard_price.txt; cat $R/guard_price.txt
T=$R/qr_device_timing_table.txt
echo "# qr_device timing table (this build, fp64, MI355X) at the nominal sizes of the reference's timing.txt (qr.cu:790-806); rocSOLVER lines: profiles/r05_qr_device_timing_table.txt" > $T
for mm in 256 512 1024 2048 4096 8192 16384 32768 65536 131072; do timeout -k 5 60 ./cuda-qr_amd/build/qr_device $mm 64 2>&1 | grep "MMQR ran" >> $T; done
for mm in 64 128 256 512 1024 2048 4096; do timeout -k 5 60 ./cuda-qr_amd/build/qr_device $mm $mm 2>&1 | grep "MMQR ran" >> $T; done
tail -9 $T
timeout -k 5 200 python3 devtools/tools_applyq.py 2>&1 | grep -v amdgpu.ids > $R/form_q_timing.txt; echo "applyq rc=$?"
timeout -k 5 400 python3 devtools/tools_fuzz_parity.py > $R/fuzz_parity.txt 2>&1; tail -3 $R/fuzz_parity.txt
timeout -k 5 400 python3 devtools/tools_cqr_fuzz.py > $R/cqr_fuzz_parity.txt 2>&1; tail -3 $R/cqr_fuzz_parity.txt
date +%T
