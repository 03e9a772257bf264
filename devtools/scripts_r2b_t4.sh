#!/bin/bash
R=gpurun_out/s2h; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -4 $R/tests.log
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %7dx%-6d %7.2f ms %6.2f TF panel %.1f' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0)))
"; }
run default
run gslab16 MI355XQR_GRAM_SLABS=16
run gslab8 MI355XQR_GRAM_SLABS=8
run gslab24 MI355XQR_GRAM_SLABS=24
run default_again
