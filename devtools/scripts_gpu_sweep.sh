#!/bin/bash
mkdir -p gpurun_out
( echo "# qr_device timing table (this build, fp64, MI355X) at the nominal sizes of the reference's timing.txt";
for mm in 256 512 1024 2048 4096 8192 16384 32768 65536 131072; do ./cuda-qr_amd/build/qr_device $mm 64 | grep "MMQR ran"; done
for mm in 64 128 256 512 1024 2048 4096; do ./cuda-qr_amd/build/qr_device $mm $mm | grep "MMQR ran"; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/timing_table.txt
tail -3 gpurun_out/timing_table.txt
# BASELINE configs: C2 (4096^2, nb 64), C3 block-size sweep, one C4 shard each for P = 2 and 4, one C5 shard
python devtools/tools_perf.py 4096x4096x64 16384x16384x32 16384x16384x64 16384x16384x128 16384x16384x256 131072x256x128 65536x256x128 262144x512x128 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(json.dumps({'m': d['m'], 'n': d['n'], 'nb': d['nb'], 'ms': round(d['ms'], 2), 'gflops': round(d['tflops'] * 1e3, 1)}))
" | tee gpurun_out/config_sweep.txt
