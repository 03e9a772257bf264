#!/bin/bash
for sp in "32" "32:0.3,128" "32:0.35,96" "32:0.4,64" "32:0.3,64" "32:0.45,96" "32:0.25,128"; do
  echo "SPLIT=$sp"
  MI355XQR_SPLIT="$sp" python3 devtools/tools_perf.py 16384x16384x256 2>/dev/null | cut -c1-90
done
