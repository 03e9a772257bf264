#!/bin/bash
mkdir -p gpurun_out
export BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for np in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 2951$np bench.py --gpus $np --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_dist$np.json 2> gpurun_out/bench_dist$np.err; echo "np=$np rc=$?"; tail -3 gpurun_out/bench_dist$np.err | cut -c1-300; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_dist$np.json') if l.startswith('{')][0]); print(d['n_gpus'], d['value'], d['ms_per_step'], d['accuracy'], d['config']['workload'], d['config']['collective'])"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_dist2w.json 2> gpurun_out/bench_dist2w.err; echo "weak np=2 rc=$?"; tail -2 gpurun_out/bench_dist2w.err | cut -c1-300; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_dist2w.json') if l.startswith('{')][0]); print(d['n_gpus'], d['value'], d['ms_per_step'], d['accuracy'], d['config']['workload'])"
