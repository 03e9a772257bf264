#!/bin/bash
O=gpurun_out/tnlab; mkdir -p $O; : > $O/out.txt
for v in "MI355XQR_KPIPE=1" "MI355XQR_KPIPE=0" "MI355XQR_TN_WIDE=1"; do
  echo "== $v" >> $O/out.txt
  env $v timeout -k 10 200 python3 devtools/tools_tn_lab.py 15872x256x16128 7936x256x8192 11008x256x16384 4096x256x16384 15872x512x16128 2>&1 | grep -v amdgpu.ids >> $O/out.txt || exit 1
done
cat $O/out.txt
