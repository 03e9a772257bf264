#!/bin/bash
# (record of a negative experiment: the MI355XQR_SSP knobs existed only in the experimental build, profiles/r03_side_stream_negative.txt)
O=gpurun_out/ssp; mkdir -p $O; : > $O/ab3.txt
for v in "MI355XQR_SSP=1" "MI355XQR_SSP=0" "MI355XQR_SSP_RESERVE=0" "MI355XQR_SSP_RESERVE=16" "MI355XQR_SSP=1" "MI355XQR_SSP=0"; do
  echo "== $v" >> $O/ab3.txt
  env $v timeout -k 10 300 python3 devtools/tools_perf.py 262144x512x128 131072x256x128 262144x256x128 2>&1 | grep -v amdgpu.ids | cut -c1-100 >> $O/ab3.txt || exit 1
done
cat $O/ab3.txt
./devtools/rounds/r3/r3_ssp_tl.sh
