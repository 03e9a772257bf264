#!/bin/bash
R=gpurun_out/soak; rm -rf $R; mkdir -p $R
for i in 1 2 3; do
  timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout=600 -p no:cacheprovider > $R/tests_$i.log 2>&1; echo "run $i tests rc=$? $(tail -1 $R/tests_$i.log)"
done
for i in 1 2; do
  python3 bench.py --steps 20 --warmup 3 > $R/bench_$i.json 2> $R/bench_$i.err; echo "bench $i rc=$?"; python3 -c "
import json; d=json.loads(open('$R/bench_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['accuracy'])"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
