#!/bin/bash
CFG="262144x512x128 65536x256x128 131072x256x128"
for g in 16 64 256; do echo "G=$g"; MI355XQR_TALL_COOP_G=$g python3 devtools/tools_perf.py $CFG 2>/dev/null | cut -c1-100; done
echo "coop off"; MI355XQR_TALL_COOP=0 python3 devtools/tools_perf.py $CFG 2>/dev/null | cut -c1-100
