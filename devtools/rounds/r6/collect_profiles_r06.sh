#!/bin/bash
# copies the summaries of the last evidence run (devtools/rounds/r6/scripts_r6_evidence.sh, scripts_r6_pmc.sh) from gpurun_out/ into profiles/
P=gpurun_out/prof_r06; M=gpurun_out/pmc_r06; O=profiles
cp $P/baseline_config_sweep.txt $O/r06_baseline_config_sweep.txt
for w in c3 tsqr c2; do
  cp $P/bench_$w.json $O/r06_bench_${w}_line.json
  cp $P/bench_${w}_under_rocprof.json $O/r06_bench_${w}_under_rocprof.json
  cp $P/$w/bench_kernel_stats.csv $O/r06_bench_${w}_kernel_stats.csv
done
cp $P/bench_tsqr_cond1e9.json $O/r06_bench_tsqr_cond1e9_line.json
cp $P/bench_c3_trace_summary.txt $O/r06_bench_c3_trace_summary.txt
cp $P/c3_gantt.txt $O/r06_c3_schedule_gantt.txt
cp $P/panel_fused_perf.txt $O/r06_panel_fused_perf.txt
[ -s $P/panel_fused_stamps.txt ] && cp $P/panel_fused_stamps.txt $O/r06_panel_fused_stamps.txt
cp $P/cqr_kernel_times.txt $O/r06_cqr_kernel_times.txt
cp $P/tsqr_rank_step_latency.txt $O/r06_tsqr_rank_step_latency.txt
cp $P/guard_price.txt $O/r06_guard_price.txt
cp $P/form_q_timing.txt $O/r06_form_q_timing.txt
cp $P/qr_device_timing_table.txt $O/r06_qr_device_timing_table.txt
cp $P/fuzz_parity.txt $O/r06_fuzz_parity.txt
cp $P/cqr_fuzz_parity.txt $O/r06_cqr_fuzz_parity.txt
[ -f $M/pmc_traffic.json ] && cp $M/pmc_traffic.json $O/r06_pmc_traffic.json
[ -f $M/pmc_mfma_lds_util.txt ] && cp $M/pmc_mfma_lds_util.txt $O/r06_pmc_mfma_lds_util.txt
[ -f $M/tsqr_total_traffic.json ] && cp $M/tsqr_total_traffic.json $O/r06_pmc_tsqr_total_traffic.json
[ -f $M/panel_kernels_hbm.txt ] && cp $M/panel_kernels_hbm.txt $O/r06_pmc_panel_kernels_hbm.txt
[ -f $M/pmc_panel_hbm.json ] && cp $M/pmc_panel_hbm.json $O/r06_pmc_panel_hbm.json
ls $O | grep -c r06_
