#!/bin/bash
# libmi355xqr_cqstamps.so: the library with phase stamps in the one-workgroup kernels of qr_panel_cqr.hip (devtools/tools_cqr_debug.py mk w stamps)
cd "$(dirname "$0")/../cuda-qr_amd" && make -s && hipcc --offload-arch=gfx950 -O3 -fPIC -DCQ_STAMPS -c csrc/qr_panel_cqr.hip -o build/qr_panel_cqr_stamps.o && \
hipcc --offload-arch=gfx950 -shared -fPIC -o libmi355xqr_cqstamps.so build/qr_kernels.o build/qr_panel_tsqr.o build/qr_gemm_nt.o build/qr_leaf_fused.o build/qr_panel_fused.o build/qr_panel_cqr_stamps.o build/qr_factor32_dbg.o build/qr_legacy.o build/qr_comm.o build/qr_host.o -lpthread -ldl
