#!/bin/bash
run() { echo "== $*"; env "$@" timeout 300 python devtools/tools_fuzz_one.py 6144 6144 128 6 2>&1 | grep "grade"; }
run A=1
run MI355XQR_EARLY_NEXT=0
run MI355XQR_BALANCE=0
run MI355XQR_LOOKAHEAD=0
run MI355XQR_PANEL=tsqr
echo "== grade 0 (plain uniform)"; timeout 300 python devtools/tools_fuzz_one.py 6144 6144 128 0 2>&1 | grep grade
echo "== grade 3"; timeout 300 python devtools/tools_fuzz_one.py 6144 6144 128 3 2>&1 | grep grade
echo "== 4096 grade 6"; timeout 300 python devtools/tools_fuzz_one.py 4096 4096 128 6 2>&1 | grep grade
