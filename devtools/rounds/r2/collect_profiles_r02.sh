#!/bin/bash
# copies the summaries of the last evidence run (devtools/rounds/r2/scripts_r2_all_evidence.sh, scripts_r2_tl.sh 3) from gpurun_out/ into profiles/
P=gpurun_out/prof_r02; M=gpurun_out/pmc_r02; L=gpurun_out/pmc_panel_r02; O=profiles
cp $P/baseline_config_sweep.txt $O/r02_baseline_config_sweep.txt
for w in c3 tsqr c2; do
  cp $P/bench_$w.json $O/r02_bench_${w}_line.json
  cp $P/bench_${w}_under_rocprof.json $O/r02_bench_${w}_under_rocprof.json
  cp $P/$w/bench_kernel_stats.csv $O/r02_bench_${w}_kernel_stats.csv
done
cp $P/bench_c3_trace_summary.txt $O/r02_bench_c3_trace_summary.txt
cp $P/c3_gantt.txt $O/r02_c3_schedule_gantt.txt
cp $P/comparator_rocsolver.txt $O/r02_comparator_rocsolver.txt
cp $P/form_q_timing.txt $O/r02_form_q_timing.txt
cp $P/qr_device_timing_table.txt $O/r02_qr_device_timing_table.txt
cp $M/FETCH_SIZE_summary.txt $O/r02_pmc_FETCH_SIZE_summary.txt
cp $M/WRITE_SIZE_summary.txt $O/r02_pmc_WRITE_SIZE_summary.txt
cp $M/driver_shapes.json $O/r02_pmc_driver_shapes.json
cp $M/r02_pmc_traffic.json $O/r02_pmc_traffic.json
( for c in MfmaUtil LdsUtil LdsBankConflict; do echo "== rocprofv3 --pmc $c (one pass), summed per kernel: total over dispatches, average per dispatch"; grep "gemm_\|dispatches" $M/${c}_summary.txt | head -12; done ) > $O/r02_pmc_mfma_lds_util.txt
cp $L/cholqr_hbm_summary.txt $O/r02_pmc_panel_hbm_cholqr2.txt
cp $L/tsqr_hbm_summary.txt $O/r02_pmc_panel_hbm_tsqr.txt
[ -f gpurun_out/tl3/timeline_late.txt ] && cp gpurun_out/tl3/timeline_late.txt $O/r02_c3_panel_stream_timeline.txt
[ -f gpurun_out/tl3/timeline_ts.txt ] && cp gpurun_out/tl3/timeline_ts.txt $O/r02_tsqr_shard_timeline.txt
ls -la $O | grep r02_ | wc -l
python3 devtools/tools_pmc_panel_json.py $L/cholqr_hbm_summary.txt "$1" > $O/r02_pmc_panel_hbm.json
