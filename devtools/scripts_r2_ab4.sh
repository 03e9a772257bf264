#!/bin/bash
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  for rep in 1 2; do
  env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $BENCHARGS > gpurun_out/r2_ab_$name.json 2> gpurun_out/r2_ab_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2_ab_$name.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-28s ms/step %7.2f  panel ms %6.2f  nn TF %5.2f (frac %.3f)  tn TF %5.2f  resid %.1e orth %.1e  shard ms %.3f" % ("$name", d["ms_per_step"], r.get("panel_ms_per_step", 0), r["achieved"], r["frac"], (r.get("companion_tn") or {}).get("achieved") or 0, d["accuracy"]["resid"], d["accuracy"]["orth"], (d.get("weak_scaling_base_1gpu") or {}).get("ms_per_step", 0)))
except Exception as e:
    print("$name parse failed", e); print(open("gpurun_out/r2_ab_$name.err").read()[-600:])
PY
  done
}
run d64_auto
run d64_auto_nt MI355XQR_UPDATE=2
run d32_upd MI355XQR_SPLIT=32 MI355XQR_NEXT=update
run d32_upd_nt MI355XQR_SPLIT=32 MI355XQR_NEXT=update MI355XQR_UPDATE=2
run d32_upd_nobal MI355XQR_SPLIT=32 MI355XQR_NEXT=update MI355XQR_BALANCE=0
