#!/bin/bash
# balance / early-slice retuning after the TN product got faster (16384^2, tools_perf best-of-2)
O=gpurun_out/bal2; mkdir -p $O; : > $O/out.txt
for v in "X=0" "MI355XQR_BALANCE=7.04,51.5,1.1,0.6" "MI355XQR_BALANCE=7.04,51.5,0.9,0.6" "MI355XQR_BALANCE=7.04,51.5,1.3,0.6" "MI355XQR_BALANCE=7.5,51.5,1.1,0.6" "MI355XQR_BALANCE=6.6,51.5,1.1,0.6" "MI355XQR_EARLY_W1=3072" "MI355XQR_EARLY_W1=5120" "X=0"; do
  echo "== $v" >> $O/out.txt
  env $v timeout -k 10 200 python3 devtools/tools_perf.py 16384x16384x256 2>&1 | grep -v amdgpu.ids | cut -c1-100 >> $O/out.txt || exit 1
done
cat $O/out.txt
