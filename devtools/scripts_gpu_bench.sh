#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "c3 rc=$?"; tail -3 gpurun_out/bench_c3.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c3.json')); print(d['value'], d['ms_per_step'], d['accuracy'], d['roofline']['achieved'])"
