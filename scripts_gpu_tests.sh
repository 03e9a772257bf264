#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout=600 -x > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/tests.log
