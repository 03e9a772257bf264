#!/bin/bash
# round 6, run 15: trsm_gt_kernel with its operands requested ahead: parity, kernel time under rocprofv3, C3 A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run15; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "trsm" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o tl -- python3 devtools/tools_perf.py 8192x8192x256 > $O/log.txt 2>&1
grep "trsm_gt\|slab_reduce_kernel\|gemm_nn_batch" $(find $O/p -name "*kernel_stats.csv" | head -1) | cut -c1-200
rm -rf $O/p
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'))
"; }
export CUDA_QR_AMD_LIB=lab
S="16384x16384x256 16384x16384x256 8192x8192x256 4096x4096x256"
( for i in 1 2; do
  echo "== MI355XQR_TRSM=0"; MI355XQR_TRSM=0 python3 devtools/tools_perf.py $S 2>&1 | fmt
  echo "== MI355XQR_TRSM=1"; MI355XQR_TRSM=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  done ) > $O/trsm_ab.txt 2>&1
cat $O/trsm_ab.txt
