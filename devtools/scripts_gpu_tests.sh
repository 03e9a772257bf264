#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/tests.log
