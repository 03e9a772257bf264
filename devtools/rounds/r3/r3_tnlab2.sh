#!/bin/bash
O=gpurun_out/tnlab; mkdir -p $O; : > $O/out2.txt
MI355XQR_KPIPE=4 timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_tn" > $O/pytest3.txt 2>&1 || { tail -30 $O/pytest3.txt; exit 1; }
tail -1 $O/pytest3.txt
for km in 1 8; do for kp in 3 4; do
  echo "== KMAX=$km KPIPE=$kp" >> $O/out2.txt
  MI355XQR_TN_KMAX=$km MI355XQR_KPIPE=$kp timeout -k 10 200 python3 devtools/tools_tn_lab.py 15872x256x16128 11008x256x16384 2>&1 | grep -v amdgpu.ids >> $O/out2.txt || exit 1
done; done
cat $O/out2.txt
