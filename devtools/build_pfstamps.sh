#!/bin/bash
# libmi355xqr_stamps.so: the library with phase stamps in the one-launch panel (devtools/tools_panel_fused_stamps.py)
cd "$(dirname "$0")/../cuda-qr_amd" && make -s && hipcc --offload-arch=gfx950 -O3 -fPIC -DPF_STAMPS -c csrc/qr_panel_fused.hip -o build/qr_panel_fused_stamps.o && \
hipcc --offload-arch=gfx950 -shared -fPIC -o libmi355xqr_stamps.so build/qr_kernels.o build/qr_panel_tsqr.o build/qr_gemm_nt.o build/qr_leaf_fused.o build/qr_panel_fused_stamps.o build/qr_panel_cqr.o build/qr_factor32_dbg.o build/qr_legacy.o build/qr_comm.o build/qr_host.o -lpthread -ldl
