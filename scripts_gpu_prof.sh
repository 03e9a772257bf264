#!/bin/bash
R=gpurun_out/pmc_try
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sz in 4096 8192 16384; do
  MI355XQR_PANEL_CUS=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/d$sz -o pmc -- python3 tools_pmc_driver.py $sz 128 > $R/d$sz.log 2>&1
  echo "driver $sz rc=$? sigsegv=$(grep -c SIGSEGV $R/d$sz.log) $(grep -c . $R/d$sz/pmc_counter_collection.csv 2>/dev/null)"
  f=$R/d$sz/pmc_counter_collection.csv
  [ -f $f ] && python3 tools_pmc_summary.py $f FETCH_SIZE | head -8
  rm -f $f
done
