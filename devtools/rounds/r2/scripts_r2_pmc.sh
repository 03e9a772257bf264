#!/bin/bash
# round 2 PMC passes over the wide-update GEMM pair (one counter per pass; no trace flags next to --pmc)
R=gpurun_out/pmc_r02; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MI355XQR_PANEL_CUS=0
for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil LdsUtil LdsBankConflict; do
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$ctr -o pmc -- python3 devtools/tools_pmc_driver2.py 256 > $R/${ctr}_driver.json 2> $R/$ctr.err
  f=$(find $R/$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 devtools/tools_pmc_summary.py $f $ctr > $R/${ctr}_summary.txt && rm -f $f
  grep "gemm_\|stream_copy\|diff_norm" $R/${ctr}_summary.txt | head -6
done
tail -1 $R/FETCH_SIZE_driver.json > $R/driver_shapes.json
python3 devtools/tools_pmc_traffic2.py $R/FETCH_SIZE_summary.txt $R/WRITE_SIZE_summary.txt $R/driver_shapes.json "$1" > $R/r02_pmc_traffic.json && cat $R/r02_pmc_traffic.json | head -40
