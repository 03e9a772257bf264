#!/bin/bash
# (record of a negative experiment: the MI355XQR_NB_EARLY / _TN_HALVES knobs existed only in the experimental build, profiles/r03_hybrid_nb_negative.txt)
# wide early panels (512) + nb 256 for the rest: sweep of the switch column at 16384^2 (and 12288^2, 24576^2)
O=gpurun_out/hybrid; mkdir -p $O; : > $O/sweep.txt
run() { echo "== $*" >> $O/sweep.txt; env "$@" CHECK=1 timeout -k 10 200 python3 devtools/tools_perf.py $SHAPE 2>&1 | grep -v amdgpu.ids | cut -c1-420 >> $O/sweep.txt || exit 1; }
SHAPE=16384x16384x256
run X=0
for f in 0.15 0.25 0.3 0.35 0.45; do run MI355XQR_NB_EARLY=512 MI355XQR_NB_EARLY_FRAC=$f; done
run MI355XQR_NB_EARLY=512 MI355XQR_NB_EARLY_FRAC=0.3 MI355XQR_TN_HALVES=0
SHAPE=16384x16384x512
run X=0
run MI355XQR_TN_HALVES=0
SHAPE=12288x12288x256
run X=0
run MI355XQR_NB_EARLY=512 MI355XQR_NB_EARLY_FRAC=0.25
SHAPE=24576x24576x256
run X=0
run MI355XQR_NB_EARLY=512 MI355XQR_NB_EARLY_FRAC=0.3
run MI355XQR_NB_EARLY=512 MI355XQR_NB_EARLY_FRAC=0.45
