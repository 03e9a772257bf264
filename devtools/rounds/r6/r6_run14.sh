#!/bin/bash
# round 6, run 14: 32 partial Gram matrices per load batch in the factor workgroup's sums: parity + A/B against HEAD's kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run14; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_panel_fused.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
( for i in 1 2; do
  echo "== previous kernel"; CUDA_QR_AMD_LIB=libmi355xqr_exp_prev.so PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids
  echo "== 32-wide sum batches"; CUDA_QR_AMD_LIB=lab PF_NO_GRAM=1 python3 devtools/tools_panel_fused_perf.py 0 2>&1 | grep -v amdgpu.ids
  done ) > $O/panel_fused_perf_ab.txt
cat $O/panel_fused_perf_ab.txt
