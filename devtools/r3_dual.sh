#!/bin/bash
for s in 8 4 16 2; do for k in 4 2 8 16; do
  echo -n "SLOTS=$s KMIN=$k : "
  MI355XQR_DUAL_SLOTS=$s MI355XQR_DUAL_KMIN=$k python3 devtools/tools_perf.py 262144x512x128 2>/dev/null | cut -c40-75
done; done
