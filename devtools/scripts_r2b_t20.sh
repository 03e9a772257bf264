#!/bin/bash
R=gpurun_out/s2q; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout=600 > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
