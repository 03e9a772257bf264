#!/bin/bash
# copies the summaries of the last evidence run (devtools/rounds/r5/scripts_r5_evidence.sh, scripts_r5_pmc.sh) from gpurun_out/ into profiles/
P=gpurun_out/prof_r05; M=gpurun_out/pmc_r05; O=profiles
cp $P/baseline_config_sweep.txt $O/r05_baseline_config_sweep.txt
for w in c3 tsqr c2; do
  cp $P/bench_$w.json $O/r05_bench_${w}_line.json
  cp $P/bench_${w}_under_rocprof.json $O/r05_bench_${w}_under_rocprof.json
  cp $P/$w/bench_kernel_stats.csv $O/r05_bench_${w}_kernel_stats.csv
done
cp $P/bench_c3_trace_summary.txt $O/r05_bench_c3_trace_summary.txt
cp $P/c3_gantt.txt $O/r05_c3_schedule_gantt.txt
cp $P/leaf_phase_stamps.txt $O/r05_leaf_phase_stamps.txt
cp $P/panel_fused_perf.txt $O/r05_panel_fused_perf.txt
cp $P/cqr_kernel_times.txt $O/r05_cqr_kernel_times.txt
[ -s $P/cq_stamps.txt ] && cp $P/cq_stamps.txt $O/r05_cq_stamps.txt
cp $P/tsqr_rank_step_latency.txt $O/r05_tsqr_rank_step_latency.txt
cp $P/comparator_rocsolver.txt $O/r05_comparator_rocsolver.txt
cp $P/form_q_timing.txt $O/r05_form_q_timing.txt
cp $P/qr_device_timing_table.txt $O/r05_qr_device_timing_table.txt
cp $P/fuzz_parity.txt $O/r05_fuzz_parity.txt
cp $P/cqr_fuzz_parity.txt $O/r05_cqr_fuzz_parity.txt
[ -f $M/pmc_traffic.json ] && cp $M/pmc_traffic.json $O/r05_pmc_traffic.json
[ -f $M/pmc_mfma_lds_util.txt ] && cp $M/pmc_mfma_lds_util.txt $O/r05_pmc_mfma_lds_util.txt
[ -f $M/tsqr_total_traffic.json ] && cp $M/tsqr_total_traffic.json $O/r05_pmc_tsqr_total_traffic.json
[ -f $M/panel_kernels_hbm.txt ] && cp $M/panel_kernels_hbm.txt $O/r05_pmc_panel_kernels_hbm.txt
[ -f $M/pmc_panel_hbm.json ] && cp $M/pmc_panel_hbm.json $O/r05_pmc_panel_hbm.json
ls $O | grep -c r05_
