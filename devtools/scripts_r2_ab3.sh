#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multipanel_golden.py "tests/test_gpu_qr.py::test_geqrf_applyq_ragged_shapes" "tests/test_gpu_qr.py::test_c3_16384_square_properties" -q -x --timeout=600 > gpurun_out/r2_t3.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2_t3.log
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
run nb256_64
BENCHARGS="--nb 512" run nb512_64
BENCHARGS="--nb 512" run nb512_32 MI355XQR_SPLIT=32
BENCHARGS="--nb 512" run nb512_32_upd MI355XQR_SPLIT=32 MI355XQR_NEXT=update
BENCHARGS="--nb 512" run nb512_64_nobal MI355XQR_BALANCE=0
BENCHARGS="--nb 512" run nb512_32_nobal MI355XQR_SPLIT=32 MI355XQR_BALANCE=0
MI355XQR_SPLIT=32 python devtools/tools_gantt.py 16384x16384x512 > gpurun_out/gantt_nb512_32.txt 2>&1
python devtools/tools_gantt.py 16384x16384x512 > gpurun_out/gantt_nb512_64.txt 2>&1
