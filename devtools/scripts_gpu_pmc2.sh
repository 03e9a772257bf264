#!/bin/bash
R=gpurun_out/pmc2; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in MfmaUtil LdsBankConflict LdsUtil MemUnitStalled SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES; do
  MI355XQR_PANEL_CUS=0 timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$ctr -o pmc -- python3 devtools/tools_pmc_driver.py 256 > $R/${ctr}_driver.json 2> $R/$ctr.err
  f=$(find $R/$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 devtools/tools_pmc_summary.py $f $ctr > $R/${ctr}_summary.txt && rm -f $f
  grep "gemm_" $R/${ctr}_summary.txt | head -3
done
