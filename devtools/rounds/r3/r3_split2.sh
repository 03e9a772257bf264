#!/bin/bash
for sp in "32" "32:0.47,U" "32:0.4,U" "32:0.3,U" "32:0.55,U" "32:0.2,U"; do
  echo "SPLIT=$sp"
  MI355XQR_SPLIT="$sp" python3 devtools/tools_perf.py 16384x16384x256 2>/dev/null | cut -c1-90
done
for sp in "64" "64:0.6,U" "64:0.4,U"; do
  echo "8192 SPLIT=$sp"
  MI355XQR_SPLIT="$sp" python3 devtools/tools_perf.py 8192x8192x256 2>/dev/null | cut -c1-90
done
