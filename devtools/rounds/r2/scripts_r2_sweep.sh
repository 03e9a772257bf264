#!/bin/bash
# CU-split sweep (panel CUs [: until fraction, ...]) at several square sizes; N(s) placement
mkdir -p gpurun_out
for spec in "64" "32" "32:0.47,64" "32:0.47,128" "64:0.47,128" "32:0.47,192" "64:0.3,160"; do
  for nxt in update auto; do
    echo "== SPLIT=$spec NEXT=$nxt"
    MI355XQR_SPLIT="$spec" MI355XQR_NEXT=$nxt python devtools/tools_perf.py 16384x16384x256 8192x8192x256 4096x4096x128 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %6dx%-6d nb %3d  %8.2f ms  %6.2f TF  panel %.1f ms nn %s' % (d['m'], d['n'], d['nb'], d['ms'], d['tflops'], d.get('panel',{}).get('ms',0), d.get('update_nn',{}).get('tflops')))
"
  done
done
