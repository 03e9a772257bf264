#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "tn_dual_short_and_tall" 2>&1 | tail -5
CFG="262144x512x128 131072x256x128 262144x256x128 2097152x512x128"
python3 devtools/tools_perf.py $CFG 2>/dev/null | cut -c1-100
echo wide off
MI355XQR_DUAL_WIDE=1 python3 devtools/tools_perf.py $CFG 2>/dev/null | cut -c1-100
echo wide min 32768
MI355XQR_DUAL_WIDE_MIN=32768 MI355XQR_EP_MAX_MB=32 python3 devtools/tools_perf.py 65536x256x128 131072x256x128 2>/dev/null | cut -c1-100
./devtools/rounds/r3/r3_prof.sh r3s_tall 262144x512x128 | head -8
