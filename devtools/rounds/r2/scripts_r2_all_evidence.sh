#!/bin/bash
# everything the round's profiles/ are made from, in one GPU call:  scripts_r2_all_evidence.sh <git head>
./devtools/rounds/r2/scripts_r2_evidence.sh $1 > gpurun_out/evidence.log 2>&1
./devtools/rounds/r2/scripts_r2_pmc.sh $1 > gpurun_out/pmc.log 2>&1
./devtools/rounds/r2/scripts_r2_pmc_panel.sh > gpurun_out/pmc_panel.log 2>&1
tail -40 gpurun_out/evidence.log
