#!/bin/bash
mkdir -p gpurun_out/r3e
for sp in "32" "32:0.47,64" "32:0.45,96" "32:0.4,128" "32:0.5,96" "32:0.35,96" "32:0.3,128" "32:0.45,64:0.2,128" "48" "32:0.45,80"; do
  echo "SPLIT=$sp" >> gpurun_out/r3e/split.txt
  MI355XQR_SPLIT="$sp" python3 devtools/tools_perf.py 16384x16384x256 2>/dev/null | cut -c1-90 >> gpurun_out/r3e/split.txt
done
for nb in 512; do
  echo "NB=$nb" >> gpurun_out/r3e/split.txt
  python3 devtools/tools_perf.py 16384x16384x$nb 2>/dev/null | cut -c1-300 >> gpurun_out/r3e/split.txt
  MI355XQR_SPLIT="32:0.45,96" python3 devtools/tools_perf.py 16384x16384x$nb 2>/dev/null | cut -c1-90 >> gpurun_out/r3e/split.txt
done
cat gpurun_out/r3e/split.txt
