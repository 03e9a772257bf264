#!/bin/bash
# panel / update CU split re-checked after the TN product got faster (tools_perf best-of-2)
O=gpurun_out/split4; mkdir -p $O; : > $O/out.txt
for shape in 8192x8192x256 10240x10240x256 12288x12288x256 16384x16384x256; do
for c in 32 48 64; do
  echo "== $shape MI355XQR_PANEL_CUS=$c" >> $O/out.txt
  MI355XQR_PANEL_CUS=$c timeout -k 10 200 python3 devtools/tools_perf.py $shape 2>&1 | grep -v amdgpu.ids | cut -c1-100 >> $O/out.txt || exit 1
done; done
grep -A1 "==" $O/out.txt | grep -v "^--" | paste - - | awk '{print $2, $3, $12}'
