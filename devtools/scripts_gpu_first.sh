#!/bin/bash
# first GPU contact: device facts, micro-probes, kernel unit tests, end-to-end parity
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/probe.log 2>&1
import cuda_qr_amd as q, json
print(json.dumps(q.device_info()))
for i in range(3):
    print("mfma_f64_tflops", q.probe_mfma_f64_tflops())
print("copy_gbps", q.probe_copy_gbps())
PY
cat gpurun_out/probe.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=300 > gpurun_out/kernels.log 2>&1; echo "kernels rc=$?"; tail -30 gpurun_out/kernels.log
timeout 1200 python -m pytest tests/test_gpu_qr.py -q -m gpu --timeout=600 > gpurun_out/qr.log 2>&1; echo "qr rc=$?"; tail -40 gpurun_out/qr.log
