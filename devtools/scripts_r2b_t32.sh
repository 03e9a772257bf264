#!/bin/bash
run() { name=$1; shift; env "$@" python devtools/tools_perf.py 16384x512x128 24576x512x128 32768x512x128 49152x512x128 65536x512x128 131072x512x128 131072x256x128 2>/dev/null | python -c "
import sys, json
print('%-12s' % '$name', ' '.join('%dx%d:%.3f' % (json.loads(l)['m'], json.loads(l)['n'], json.loads(l)['ms']) for l in sys.stdin))
"; }
run min20000
run min8192 MI355XQR_FUSE_NN_MIN=8192
run min12288 MI355XQR_FUSE_NN_MIN=12288
run min32768 MI355XQR_FUSE_NN_MIN=32768
run min65536 MI355XQR_FUSE_NN_MIN=65536
run off MI355XQR_FUSE_NN=0
run min20000
