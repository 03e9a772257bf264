#!/bin/bash
# round 2: schedule A/B on C3 (leaf generation, where N(s) runs, CU split)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multipanel_golden.py -q -x --timeout=600 > gpurun_out/r2_kernels.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2_kernels.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCHARGS > gpurun_out/r2_ab_$name.json 2> gpurun_out/r2_ab_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2_ab_$name.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-28s ms/step %7.2f  panel ms %6.2f  nn TF %5.2f (frac %.3f)  tn TF %5.2f  resid %.1e orth %.1e  shard ms %.3f" % ("$name", d["ms_per_step"], r.get("panel_ms_per_step", 0), r["achieved"], r["frac"], (r.get("companion_tn") or {}).get("achieved") or 0, d["accuracy"]["resid"], d["accuracy"]["orth"], (d.get("weak_scaling_base_1gpu") or {}).get("ms_per_step", 0)))
except Exception as e:
    print("$name parse failed", e); print(open("gpurun_out/r2_ab_$name.err").read()[-600:])
PY
}
run leaf1_panel64 MI355XQR_LEAF=1 MI355XQR_NEXT=panel
run leaf2_panel64 MI355XQR_LEAF=2 MI355XQR_NEXT=panel
run leaf2_upd64 MI355XQR_LEAF=2 MI355XQR_NEXT=update
run leaf2_upd48 MI355XQR_LEAF=2 MI355XQR_NEXT=update MI355XQR_PANEL_CUS=48
run leaf2_upd32 MI355XQR_LEAF=2 MI355XQR_NEXT=update MI355XQR_PANEL_CUS=32
run leaf2_upd32_nobal MI355XQR_LEAF=2 MI355XQR_NEXT=update MI355XQR_PANEL_CUS=32 MI355XQR_BALANCE=0
run leaf2_upd24 MI355XQR_LEAF=2 MI355XQR_NEXT=update MI355XQR_PANEL_CUS=24
run leaf2_upd16 MI355XQR_LEAF=2 MI355XQR_NEXT=update MI355XQR_PANEL_CUS=16
