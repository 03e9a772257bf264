#!/bin/bash
# round 6, run 1: the whole -m gpu suite on the product + lab libraries, the driver's bench line, the price of a refused panel
# (bench.py --cond), and lab A/Bs of the tail of the C3 schedule (flat chain-time model, unmasked panel stream, wide TN tiles)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?"
python bench.py --workload tsqr --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tsqr.json 2> $O/bench_tsqr.err; echo "bench tsqr rc=$?"
python bench.py --workload tsqr --cond 1e9 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_tsqr_cond1e9.json 2> $O/bench_tsqr_cond.err; echo "bench tsqr cond rc=$?"
fmt() { grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['m'], d['n'], d['nb'], 'ms %.3f' % d['ms'], 'panel', d.get('panel', {}).get('ms'), {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"; }
export CUDA_QR_AMD_LIB=lab
S="16384x16384x256 16384x16384x256 8192x8192x256 4096x4096x64"
( echo "== lab default"; python3 devtools/tools_perf.py $S 2>&1 | fmt
  for tc in 0.7 0.8 0.9 1.0; do echo "== MI355XQR_TAILTC=$tc"; MI355XQR_TAILTC=$tc python3 devtools/tools_perf.py $S 2>&1 | fmt; done
  echo "== MI355XQR_TN_WIDE=1"; MI355XQR_TN_WIDE=1 python3 devtools/tools_perf.py $S 2>&1 | fmt
  for sp in "32:0.47,U" "32:0.35,U" "32:0.25,U"; do echo "== MI355XQR_SPLIT=$sp"; MI355XQR_SPLIT=$sp python3 devtools/tools_perf.py 16384x16384x256 16384x16384x256 2>&1 | fmt; done
  echo "== MI355XQR_TAILTC=0.8 MI355XQR_SPLIT=32:0.35,U"; MI355XQR_TAILTC=0.8 MI355XQR_SPLIT=32:0.35,U python3 devtools/tools_perf.py 16384x16384x256 16384x16384x256 2>&1 | fmt
  echo "== lab default again"; python3 devtools/tools_perf.py 16384x16384x256 16384x16384x256 2>&1 | fmt
) > $O/tail_ab.txt 2>&1
cat $O/tail_ab.txt
