#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x16384x256 12288x12288x256 8192x8192x256 4096x4096x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-28s %7dx%-6d %7.2f ms %6.2f TF panel %.1f tn %s nn %s' % ('$name', d['m'], d['n'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0), d.get('vta_tn',{}).get('tflops'), d.get('update_nn',{}).get('tflops')))
"; }
run default
run spread MI355XQR_PANEL_SPREAD=1
run spread_c64 MI355XQR_PANEL_SPREAD=1 MI355XQR_SPLIT=64
run spread_c16 MI355XQR_PANEL_SPREAD=1 MI355XQR_SPLIT=16
run spread_c48 MI355XQR_PANEL_SPREAD=1 MI355XQR_SPLIT=48
run default
