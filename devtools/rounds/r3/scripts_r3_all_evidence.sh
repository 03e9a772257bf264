#!/bin/bash
# everything the round's profiles/ are made from, in one GPU call:  scripts_r3_all_evidence.sh <git head>
./devtools/rounds/r3/scripts_r3_evidence.sh $1 > gpurun_out/evidence.log 2>&1
./devtools/rounds/r3/scripts_r3_pmc.sh $1 > gpurun_out/pmc.log 2>&1
./devtools/rounds/r3/scripts_r3_pmc_panel.sh > gpurun_out/pmc_panel.log 2>&1
tail -40 gpurun_out/evidence.log
