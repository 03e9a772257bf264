#!/bin/bash
R=gpurun_out/s2n; rm -rf $R; mkdir -p $R
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err; echo "rc=$?"
python3 bench.py --workload tsqr --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_tsqr.json 2> $R/bench_tsqr.err; echo "rc=$?"
python3 bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_c2.json 2> $R/bench_c2.err; echo "rc=$?"
python3 - <<PY
import json
for n in ("bench_c3", "bench_tsqr", "bench_c2"):
    d = json.loads(open("$R/%s.json" % n).read().strip().splitlines()[-1]); r = d["roofline"]
    print(n, "value %.0f GF/s  ms %.2f  acc %s  roof %s %.4g %s frac %.3f launches %s panel %s" % (d["value"], d["ms_per_step"], d["accuracy"], r["bound"], r["achieved"], r["unit"], r["frac"], r["launches"], r.get("panel_ms_per_step")))
PY
tail -3 $R/bench_c3.err
