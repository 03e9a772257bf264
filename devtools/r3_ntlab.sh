#!/bin/bash
O=gpurun_out/ntlab; mkdir -p $O; : > $O/out.txt
MI355XQR_NT_IL=2 timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -1 $O/pytest.txt
for v in "MI355XQR_NT_IL=1" "MI355XQR_NT_IL=2"; do
  echo "== $v" >> $O/out.txt
  env $v timeout -k 10 100 python3 devtools/tools_nt_lab.py 16384x16128x256 16384x15872x512 8192x7936x256 4096x3840x256 2>&1 | grep -v amdgpu.ids >> $O/out.txt || exit 1
done
cat $O/out.txt
