#!/bin/bash
# stall breakdown of the wide TN product, isolated (tools_tn_lab.py), one counter per rocprofv3 pass
R=gpurun_out/tn_pmc; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for km in 1 8; do
for ctr in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA; do
  MI355XQR_TN_KMAX=$km MI355XQR_KPIPE=0 timeout 200 rocprofv3 --pmc $ctr --output-format csv -d $R/$km/$ctr -o pmc -- python3 devtools/tools_tn_lab.py 15872x256x16128 > $R/${km}_${ctr}.log 2>&1
  f=$(find $R/$km/$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 devtools/tools_pmc_summary.py $f $ctr | grep "gemm_tn_kernel" | head -1 | sed "s/^/KMAX=$km $ctr /" >> $R/summary.txt
  rm -rf $R/$km/$ctr
done
done
cat $R/summary.txt
