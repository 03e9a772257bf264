#!/bin/bash
cd $GRAFT_REPO_ROOT
cp cuda-qr_amd/libmi355xqr.so /tmp/new.so
for v in newchol_oldlu oldchol_newlu; do
  cp cuda-qr_amd/libmi355xqr_exp_$v.so cuda-qr_amd/libmi355xqr.so
  echo "== $v"; SWEEP_ONLY=4096 python3 devtools/r5_guard_sweep.py 2>&1 | grep -v amdgpu
done
cp /tmp/new.so cuda-qr_amd/libmi355xqr.so
