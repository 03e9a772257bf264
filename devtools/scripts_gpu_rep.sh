#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4; do
timeout 900 python -m pytest tests/test_gpu_qr.py -q -m gpu --timeout=600 -k "c2 or c3 or tall" > gpurun_out/tests_$i.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/tests_$i.log; grep "^E  " gpurun_out/tests_$i.log | head -5
done
