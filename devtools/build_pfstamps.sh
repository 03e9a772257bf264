#!/bin/bash
# libmi355xqr_stamps.so: the lab library with phase stamps in the one-launch panel (devtools/tools_panel_fused_stamps.py)
cd "$(dirname "$0")/../cuda-qr_amd" && make -s lab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DQR_LAB -DPF_STAMPS -c csrc/qr_panel_fused.hip -o build/lab/qr_panel_fused_stamps.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmi355xqr_stamps.so build/lab/qr_kernels.o build/lab/qr_panel_tsqr.o build/lab/qr_gemm_nt.o build/lab/qr_leaf_fused.o build/lab/qr_panel_fused_stamps.o build/lab/qr_panel_cqr.o build/lab/qr_factor32_dbg.o build/lab/qr_legacy.o build/lab/qr_comm.o build/lab/qr_host.o -lpthread -ldl
