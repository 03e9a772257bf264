#!/bin/bash
# final evidence of round 2 (second half): scripts_r2b_final.sh <git head>
./devtools/rounds/r2/scripts_r2_all_evidence.sh $1
./devtools/rounds/r2/scripts_r2_tl.sh 3 > gpurun_out/tl3.log 2>&1
python3 devtools/tools_records.py 16384x16384x256 30 46 2>/dev/null > gpurun_out/rec_final.txt
tail -3 gpurun_out/prof_r02/tests.log
