#!/bin/bash
# round 3, first GPU call: full GPU suite + default bench + baseline timings incl. the stacked-R shapes
set -o pipefail
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3a/tests.log
tail -5 gpurun_out/r3a/tests.log
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r3a/bench_c3.json 2> gpurun_out/r3a/bench_c3.err; echo "bench rc=$?"
python3 devtools/tools_perf.py 4096x512x128 4096x512x64 4096x512x256 4096x512x512 1024x256x128 1024x256x64 4096x4096x64 8192x8192x256 262144x512x128 65536x256x128 131072x256x128 > gpurun_out/r3a/perf.txt 2>&1
MI355XQR_PANEL=tsqr python3 devtools/tools_perf.py 4096x512x128 4096x512x512 > gpurun_out/r3a/perf_tsqr.txt 2>&1
cat gpurun_out/r3a/perf.txt gpurun_out/r3a/perf_tsqr.txt | cut -c1-260
