#!/bin/bash
# the three bench lines and their rocprofv3 kernel stats again (after a change to bench.py or a kernel): updates gpurun_out/prof_r04 in place
R=gpurun_out/prof_r04; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 10 --warmup 3 > $R/bench_c3.json 2> $R/bench_c3.err; echo "bench c3 rc=$?"
python3 bench.py --workload tsqr --steps 10 --warmup 3 > $R/bench_tsqr.json 2> $R/bench_tsqr.err; echo "bench tsqr rc=$?"
python3 bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_c2.json 2> $R/bench_c2.err; echo "bench c2 rc=$?"
for w in c3 tsqr c2; do
  rm -rf $R/$w
  wl=""; [ $w != c3 ] && wl="--workload $w"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$w -o bench -- python3 bench.py $wl --steps 3 --warmup 1 --no-cpu-baseline > $R/bench_${w}_under_rocprof.json 2> $R/$w.err
  [ $w = c3 ] && python3 devtools/tools_trace_summary.py $R/c3/bench_kernel_trace.csv > $R/bench_c3_trace_summary.txt
  rm -f $R/$w/bench_kernel_trace.csv
done
python3 - <<PY
import json
for n in ("bench_c3", "bench_tsqr", "bench_c2", "bench_c3_under_rocprof"):
    d = json.loads(open("$R/%s.json" % n).read().strip().splitlines()[-1]); r = d["roofline"]
    print(n, "value %.0f GF/s  ms %.2f  roof %s %.3g %s frac %.3f traffic %s" % (d["value"], d["ms_per_step"], r["bound"], r["achieved"], r["unit"], r["frac"], r.get("traffic")))
PY
bash devtools/r4_cqr_e2.sh 262144 128 > $R/cqr_kernel_times.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $R/bench_c3_driver_settings.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$R/bench_c3_driver_settings.json').read().strip().splitlines()[-1]); print('driver settings (20/5): ms %.2f value %.0f' % (d['ms_per_step'], d['value']))"
