#!/bin/bash
# round-4 PMC passes (one counter per pass; no trace flags next to --pmc):
#  (1) whole-factorisation HBM traffic of the 262144 x 512 shard: default routes, and with the full-width tall panel forced on;
#  (2) HBM traffic of the one-launch panel and of one full-width tall panel, per kernel.
R=gpurun_out/pmc_r04; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for tag in default cqr; do
  [ $tag = cqr ] && export MI355XQR_CQR_MIN_ROWS=32768 || unset MI355XQR_CQR_MIN_ROWS
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/total_${tag}_$ctr -o pmc -- python3 devtools/tools_one.py 262144x512x128 > $R/total_${tag}_$ctr.log 2> $R/total_${tag}_$ctr.err
  done
  f1=$(find $R/total_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $R/total_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python3 devtools/tools_pmc_total.py $f1 $f2 262144 512 2 > $R/tsqr_total_traffic_$tag.json; rm -f $f1 $f2
  head -8 $R/tsqr_total_traffic_$tag.json
done
unset MI355XQR_CQR_MIN_ROWS
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/cqr_$ctr -o pmc -- python3 devtools/tools_cqr_perf.py 262144 128 0 > $R/cqr_$ctr.log 2> $R/cqr_$ctr.err
  f=$(find $R/cqr_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 devtools/tools_pmc_summary.py $f $ctr > $R/cqr_${ctr}_summary.txt && rm -f $f
  PF_NO_GRAM=1 timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/pf_$ctr -o pmc -- python3 devtools/tools_panel_fused_perf.py > $R/pf_$ctr.log 2> $R/pf_$ctr.err
  f=$(find $R/pf_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 devtools/tools_pmc_summary.py $f $ctr > $R/pf_${ctr}_summary.txt && rm -f $f
done
( echo "# one full-width tall panel, 262144 x 128 (tools_cqr_perf.py: 6 calls), raw counters in KiB; bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950)"; grep "cqr_\|dispatches" $R/cqr_FETCH_SIZE_summary.txt | head -10; grep "cqr_\|dispatches" $R/cqr_WRITE_SIZE_summary.txt | head -10
  echo; echo "# one-launch panel (tools_panel_fused_perf.py: every shape of its table, 21 launches each)"; grep "panel_fused\|dispatches" $R/pf_FETCH_SIZE_summary.txt | head -4; grep "panel_fused\|dispatches" $R/pf_WRITE_SIZE_summary.txt | head -4 ) > $R/panel_kernels_hbm.txt
cat $R/panel_kernels_hbm.txt
