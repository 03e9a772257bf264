#!/bin/bash
mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe.log
import cuda_qr_amd as q, json
print(json.dumps(q.device_info()))
for i in range(2):
    print(json.dumps(q.probe_mfma_f64_tflops()))
print("copy_gbps", q.probe_copy_gbps())
PY
