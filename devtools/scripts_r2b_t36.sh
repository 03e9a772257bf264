#!/bin/bash
python devtools/tools_leaf_stamps.py 262144 65536 2>&1 | grep -v amdgpu
