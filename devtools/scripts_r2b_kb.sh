#!/bin/bash
R=gpurun_out/s2e; rm -rf $R; mkdir -p $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=300 -k "leaf_update_gram or cholqr" > $R/t1.log 2>&1; echo "kernel tests rc=$?"; tail -4 $R/t1.log
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -4 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 262144x512x128 262144x256x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %6dx%-6d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run fused
run nofuse MI355XQR_FUSE_NN=0
run fused_short MI355XQR_FUSE_NN_MAX=32768
run fused_gy0 MI355XQR_FUSE_NN_GY=0
