#!/bin/bash
# the last third of devtools/scripts_r5_evidence.sh alone (timing table, form-Q, comparator, fuzz parity): same output directory
R=gpurun_out/prof_r05; mkdir -p $R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# (every run appends as it finishes: a tool that stops writing for minutes is taken to be hung by the GPU pool; the vendor comparator's
# first call has repeatedly not returned inside long sessions, hence the timeouts)
# the vendor comparator first: its first call pages rocBLAS's code objects in (minutes on a fresh box, during which nothing is printed:
# hence the heartbeat), after which `qr_device --compare` answers in seconds
( while sleep 60; do date +%T >> $R/heartbeat.txt; done ) & HB=$!
timeout -k 5 600 python3 devtools/tools_comparator.py 2>&1 | grep -v amdgpu.ids > $R/comparator_rocsolver.txt; tail -5 $R/comparator_rocsolver.txt
T=$R/qr_device_timing_table.txt
echo "# qr_device timing table (this build, fp64, MI355X) at the nominal sizes of the reference's timing.txt; --compare adds the rocSOLVER line (qr.cu:790-806)" > $T
for mm in 256 512 1024 2048 4096 8192 16384 32768 65536 131072; do timeout -k 5 60 ./cuda-qr_amd/build/qr_device $mm 64 2>&1 | grep "MMQR ran" >> $T; done
for mm in 64 128 256 512 1024 2048 4096; do timeout -k 5 60 ./cuda-qr_amd/build/qr_device $mm $mm 2>&1 | grep "MMQR ran" >> $T; done
for s in "8192 8192" "16384 16384" "262144 512"; do timeout -k 5 300 stdbuf -oL ./cuda-qr_amd/build/qr_device $s --compare 2>&1 | grep "ran QR" >> $T; echo "compare $s rc=$?"; done
kill $HB
tail -9 $T
timeout -k 5 200 python3 devtools/tools_applyq.py 2>&1 | grep -v amdgpu.ids > $R/form_q_timing.txt; echo "applyq rc=$?"
timeout -k 5 400 python3 devtools/tools_fuzz_parity.py > $R/fuzz_parity.txt 2>&1; tail -3 $R/fuzz_parity.txt
timeout -k 5 400 python3 devtools/tools_cqr_fuzz.py > $R/cqr_fuzz_parity.txt 2>&1; tail -3 $R/cqr_fuzz_parity.txt
date +%T
