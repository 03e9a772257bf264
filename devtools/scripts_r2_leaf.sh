#!/bin/bash
# round 2: leaf generation A/B (MI355XQR_LEAF=1 old 7-launch CholeskyQR2 leaf, =2 fused 4-launch leaf)
mkdir -p gpurun_out
date +%T
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout=600 > gpurun_out/r2_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -5 gpurun_out/r2_kernels.log
date +%T
for leaf in 1 2; do
  MI355XQR_LEAF=$leaf timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_c3_leaf$leaf.json 2> gpurun_out/r2_c3_leaf$leaf.err; echo "c3 leaf=$leaf rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2_c3_leaf$leaf.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("C3 leaf=$leaf ms/step %.2f  panel ms %.2f  nn TF %.2f  tn TF %.2f  resid %.2e orth %.2e  shard ms %.3f" % (d["ms_per_step"], r["panel_ms_per_step"], r["achieved"], r["companion_tn"]["achieved"], d["accuracy"]["resid"], d["accuracy"]["orth"], d["weak_scaling_base_1gpu"]["ms_per_step"]))
except Exception as e:
    print("parse failed", e)
PY
  date +%T
done
