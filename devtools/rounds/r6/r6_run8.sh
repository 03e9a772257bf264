#!/bin/bash
# round 6, run 8: the whole -m gpu suite with the round's defaults, the bench lines, one rank's TSQR step with / without the halved last block
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run8; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputests.log
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?"
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
python bench.py --workload tsqr --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tsqr.json 2> $O/bench_tsqr.err; echo "bench tsqr rc=$?"
python bench.py --workload tsqr --cond 1e9 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_tsqr_cond1e9.json 2> $O/bench_tsqr_cond.err; echo "bench tsqr cond rc=$?"
python3 - <<'PY'
import json
for f in ("bench_c3", "bench_c2", "bench_tsqr", "bench_tsqr_cond1e9"):
    try:
        d = json.loads(open("gpurun_out/r6_run8/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms %.3f" % d["ms_per_step"], "GF %.0f" % d["value"], d["accuracy"], d.get("panel_routes"), "frac", (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
export CUDA_QR_AMD_LIB=lab
( for h in 3072 0; do echo "== MI355XQR_TSQR_HALVES=$h"; MI355XQR_TSQR_HALVES=$h python3 devtools/tools_tsqr_latency.py 262144x512x8x128 65536x256x4x128 131072x256x2x128 2>&1 | grep -v amdgpu.ids; done ) > $O/tsqr_rank_latency.txt
cat $O/tsqr_rank_latency.txt
unset CUDA_QR_AMD_LIB
python3 devtools/tools_gantt.py 16384x16384x256 2>&1 | grep -v amdgpu.ids > $O/gantt.txt; head -2 $O/gantt.txt
