#!/bin/bash
for b in "" "7.04,51.5,0.6,0.72" "7.04,51.5,0.7,0.7" "7.04,51.5,0.5,0.8" "7.04,51.5,0.8,0.6" "7.04,51.5,0.4,0.7" "7.04,51.5,0.9,0.6" "7.04,51.5,0.3,0.6" "7.04,49,0.6,0.72" "7.7,51.5,0.6,0.72"; do
  echo -n "BALANCE=$b : "
  if [ -z "$b" ]; then python3 devtools/tools_perf.py 16384x16384x256 2>/dev/null | cut -c40-75; else MI355XQR_BALANCE="$b" python3 devtools/tools_perf.py 16384x16384x256 2>/dev/null | cut -c40-75; fi
done
