#!/bin/bash
# round-5 PMC passes (one counter per pass; no trace flags next to --pmc):
#  (1) the trailing-update GEMM pair of C3 (whole chip: rocprofv3 --pmc crashes on CU-masked streams on this pool): HBM bytes, MfmaUtil, LDS;
#  (2) whole-factorisation HBM traffic of the 262144 x 512 shard;
#  (3) HBM traffic of one full-width tall panel and of the one-launch panel, per kernel.   Usage: scripts_r5_pmc.sh <git head>
R=gpurun_out/pmc_r05; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MI355XQR_PANEL_CUS=0
for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil LdsUtil LdsBankConflict; do
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$ctr -o pmc -- python3 devtools/tools_pmc_driver2.py 256 > $R/${ctr}_driver.json 2> $R/$ctr.err
  f=$(find $R/$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 devtools/tools_pmc_summary.py $f $ctr > $R/${ctr}_summary.txt && rm -f $f
  grep "gemm_\|stream_copy\|diff_norm" $R/${ctr}_summary.txt | head -6
done
tail -1 $R/FETCH_SIZE_driver.json > $R/driver_shapes.json
python3 devtools/tools_pmc_traffic2.py $R/FETCH_SIZE_summary.txt $R/WRITE_SIZE_summary.txt $R/driver_shapes.json "$1" > $R/pmc_traffic.json && head -40 $R/pmc_traffic.json
( echo "# MfmaUtil / LdsUtil / LdsBankConflict of the trailing-update GEMM pair (devtools/tools_pmc_driver2.py 256, whole chip), one rocprofv3 --pmc pass each";
  for ctr in MfmaUtil LdsUtil LdsBankConflict; do echo "## $ctr"; grep "gemm_" $R/${ctr}_summary.txt | head -4; done ) > $R/pmc_mfma_lds_util.txt
unset MI355XQR_PANEL_CUS
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/total_$ctr -o pmc -- python3 devtools/tools_one.py 262144x512x128 > $R/total_$ctr.log 2> $R/total_$ctr.err
done
f1=$(find $R/total_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $R/total_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 devtools/tools_pmc_total.py $f1 $f2 262144 512 2 > $R/tsqr_total_traffic.json; rm -f $f1 $f2
head -12 $R/tsqr_total_traffic.json
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
python3 devtools/tools_pmc_panel_cqr_json.py $R/cqr_FETCH_SIZE_summary.txt $R/cqr_WRITE_SIZE_summary.txt "$1" > $R/pmc_panel_hbm.json && head -20 $R/pmc_panel_hbm.json
