#!/bin/bash
# copies the summaries of the last evidence run (devtools/rounds/r4/scripts_r4_evidence.sh, scripts_r4_pmc.sh) from gpurun_out/ into profiles/
P=gpurun_out/prof_r04; M=gpurun_out/pmc_r04; O=profiles
cp $P/baseline_config_sweep.txt $O/r04_baseline_config_sweep.txt
for w in c3 tsqr c2; do
  cp $P/bench_$w.json $O/r04_bench_${w}_line.json
  cp $P/bench_${w}_under_rocprof.json $O/r04_bench_${w}_under_rocprof.json
  cp $P/$w/bench_kernel_stats.csv $O/r04_bench_${w}_kernel_stats.csv
done
cp $P/bench_c3_trace_summary.txt $O/r04_bench_c3_trace_summary.txt
cp $P/c3_gantt.txt $O/r04_c3_schedule_gantt.txt
cp $P/c3_panel_stream_timeline.txt $O/r04_c3_panel_stream_timeline.txt
cp $P/panel_fused_perf.txt $O/r04_panel_fused_perf.txt
cp $P/fused_ab.txt $O/r04_fused_ab.txt
cp $P/cu_split_tail.txt $O/r04_cu_split_tail.txt
cp $P/cqr_kernel_times.txt $O/r04_cqr_kernel_times.txt
cp $P/cqr_vs_leaves.txt $O/r04_cqr_vs_leaves.txt
cp $P/comparator_rocsolver.txt $O/r04_comparator_rocsolver.txt
cp $P/form_q_timing.txt $O/r04_form_q_timing.txt
cp $P/qr_device_timing_table.txt $O/r04_qr_device_timing_table.txt
cp $P/fuzz_parity.txt $O/r04_fuzz_parity.txt
[ -f $M/tsqr_total_traffic_default.json ] && cp $M/tsqr_total_traffic_default.json $O/r04_pmc_tsqr_total_traffic.json
[ -f $M/panel_kernels_hbm.txt ] && cp $M/panel_kernels_hbm.txt $O/r04_pmc_panel_kernels_hbm.txt
ls $O | grep -c r04_
