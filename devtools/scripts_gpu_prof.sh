#!/bin/bash
R=gpurun_out/prof_r01
rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_c3 -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err
python3 devtools/tools_trace_summary.py $R/bench_c3/bench_kernel_trace.csv > $R/bench_c3_trace_summary.txt; rm -f $R/bench_c3/bench_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_tsqr -o bench -- python3 bench.py --workload tsqr --steps 3 --warmup 1 --no-cpu-baseline > $R/bench_tsqr.json 2> $R/bench_tsqr.err
rm -f $R/bench_tsqr/bench_kernel_trace.csv
for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil LdsUtil LdsBankConflict; do
  MI355XQR_PANEL_CUS=0 rocprofv3 --pmc $ctr --output-format csv -d $R/pmc_$ctr -o pmc -- python3 devtools/tools_pmc_driver.py 256 > $R/pmc_${ctr}_driver.json 2> $R/pmc_$ctr.err
  f=$(find $R/pmc_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 devtools/tools_pmc_summary.py $f $ctr > $R/pmc_${ctr}_summary.txt && rm -f $f
  head -8 $R/pmc_${ctr}_summary.txt
done
python -c "
import json; d=json.load(open('$R/bench_c3.json')); print('C3', d['value'], d['ms_per_step'], d['accuracy'], d['roofline']['achieved'], d['roofline']['companion_tn']['achieved'])
d=json.load(open('$R/bench_tsqr.json')); print('TSQR', d['value'], d['ms_per_step'], d['accuracy'])"
head -12 $R/bench_c3/bench_kernel_stats.csv | cut -c1-150
python3 devtools/tools_probe_mfma.py 2>&1 | grep -v amdgpu.ids > $R/probe_mfma_vs_cus.txt
python3 devtools/tools_probe_gemmk.py 2>&1 | grep -v amdgpu.ids | grep gemm_nn > $R/probe_gemm_vs_k.txt
python3 devtools/tools_clock_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "Exception\|Traceback\|File\|Attribute" > $R/probe_clock_power.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids > $R/device_probes.txt
import cuda_qr_amd as q, json
print(json.dumps(q.device_info()))
for i in range(2):
    print(json.dumps(q.probe_mfma_f64_tflops()))
print("copy_gbps", q.probe_copy_gbps())
PY
cat $R/device_probes.txt
