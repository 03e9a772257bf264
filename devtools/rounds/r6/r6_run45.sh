#!/bin/bash
# round 6, run 45: host-pointer entry points pad heights that are not multiples of 16 with zero rows on the device: tests, then qr_device timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run45; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_qr.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
( for s in "5000 5000" "5001 5001" "5002 5002" "8191 8191" "8192 8192" "10001 3001" "10000 3000" "777 555" "100001 500"; do timeout -k 5 90 ./cuda-qr_amd/build/qr_device $s 2>&1 | grep "MMQR ran"; done ) | tee $O/qr_device_odd.txt
