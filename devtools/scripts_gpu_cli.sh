#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_qr.py -q -m gpu -k "qr_device or c_caller" --timeout=300 2>&1 | tail -3
( echo "# qr_device timing table (this build, fp64, MI355X) at the nominal sizes of the reference's timing.txt"; 
for mm in 256 512 1024 2048 4096 8192 16384 32768 65536 131072; do ./cuda-qr_amd/build/qr_device $mm 64 | grep "MMQR ran"; done
for mm in 64 128 256 512 1024 2048 4096; do ./cuda-qr_amd/build/qr_device $mm $mm | grep "MMQR ran"; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/timing_table.txt
