#!/bin/bash
# round 6, run 25: parity fuzz on the shapes next to the thresholds of the round-6 block-size / look-ahead / CU-split rules (library defaults, nb = 0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run25; mkdir -p $O
timeout -k 5 500 python3 devtools/tools_fuzz_parity.py 6 edges6 2>&1 | grep -v amdgpu.ids > $O/fuzz_edges6.txt; rc=$?; cat $O/fuzz_edges6.txt; echo rc=$rc
