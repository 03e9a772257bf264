"""time the one-launch panel (qrd_panel_fused_rows) on a few panel shapes: us per launch, us per leaf.
python devtools/tools_panel_fused_perf.py [rows per row workgroup: 0 = library's choice | 128 | 256]   (PF_NO_GRAM=1: without the Gram blocks, as the plans call it)"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import cuda_qr_amd as qr

lib = qr.lib
qr.check(lib.qrd_init(), "init")
f = lib.qrd_panel_fused_rows
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
              C.c_void_p, C.POINTER(C.c_uint), C.c_void_p, C.c_int]
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.qrd_panel_fused_ws_doubles.restype = C.c_size_t
ws = torch.zeros(int(lib.qrd_panel_fused_ws_doubles()), dtype=torch.float64, device="cuda")
epoch = C.c_uint(0)
status = torch.zeros(4, dtype=torch.int32, device="cuda")
shapes = [(256, 64), (512, 64), (1024, 64), (2048, 64), (3072, 64), (4096, 64), (1024, 128), (2048, 128), (4096, 128), (8192, 128), (2048, 256), (4096, 256), (8192, 256)]
if __import__('os').environ.get("PF_TALL"):       # panels of more than 8192 rows (MI355XQR_PF_MAX_ROWS in the lab build)
    shapes = [(8192, 128), (12288, 128), (16384, 128), (8192, 256), (12288, 256), (16128, 256), (16384, 256), (12288, 64), (16384, 64)]
print(f"rows per row workgroup: {ROWS or 'library choice'}; Gram blocks: {'no' if __import__('os').environ.get('PF_NO_GRAM') else 'yes'}")
reps = 20
for mk, wh in shapes:
    rng = np.random.default_rng(1)
    P = torch.from_numpy(np.ascontiguousarray(rng.random((wh, mk)))).cuda()
    bufs = [P.clone() for _ in range(reps + 3)]
    V = torch.zeros((wh, mk), dtype=torch.float64, device="cuda")
    T = torch.zeros((wh, wh), dtype=torch.float64, device="cuda")
    G = torch.zeros((wh, wh), dtype=torch.float64, device="cuda")
    tau = torch.zeros(wh, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    import os
    gp = None if os.environ.get("PF_NO_GRAM") else G.data_ptr()
    def go(b):
        rc = f(None, b.data_ptr(), mk, mk, wh, tau.data_ptr(), T.data_ptr(), wh, V.data_ptr(), mk, gp, wh, ws.data_ptr(),
               C.byref(epoch), status.data_ptr(), ROWS)
        assert rc == 0, rc
    for i in range(3):
        go(bufs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        go(bufs[3 + i])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print(f"{mk:6d} x {wh:3d}: {us:8.1f} us per panel, {us / (wh // 32):6.1f} us per leaf   status {status.cpu().numpy()[:2]}")
