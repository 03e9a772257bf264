#!/bin/bash
run() { echo "== $*"; env "$@" timeout 120 python devtools/tools_fuzz_one.py 2048 2048 128 6 2>&1 | grep "grade"; }
run A=1
run MI355XQR_PANEL=tsqr
run MI355XQR_LEAF=2
run MI355XQR_LEAF_HALFWG=0
run MI355XQR_LOOKAHEAD=0
timeout 120 python devtools/tools_fuzz_one.py 2048 2048 128 3 2>&1 | grep grade
timeout 120 python devtools/tools_fuzz_one.py 2048 512 128 6 2>&1 | grep grade
timeout 120 python devtools/tools_fuzz_one.py 1024 1024 128 6 2>&1 | grep grade
timeout 120 python devtools/tools_fuzz_one.py 1024 256 128 6 2>&1 | grep grade
