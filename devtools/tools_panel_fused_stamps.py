"""phase stamps of the one-launch panel (development library built with -DPF_STAMPS): where a leaf's time goes, workgroup 0"""
import ctypes as C
import os
import sys

import numpy as np
import torch

lib = C.CDLL(os.path.abspath("cuda-qr_amd/libmi355xqr_stamps.so"))
assert lib.qrd_init() == 0
f = lib.qrd_panel_fused
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
              C.c_void_p, C.POINTER(C.c_uint), C.c_void_p]
lib.qrd_panel_fused_ws_doubles.restype = C.c_size_t
lib.qrd_panel_fused_set_stamps.argtypes = [C.c_void_p]
ws = torch.zeros(int(lib.qrd_panel_fused_ws_doubles()), dtype=torch.float64, device="cuda")
stamps = torch.zeros(8 * 32, dtype=torch.int64, device="cuda")
lib.qrd_panel_fused_set_stamps(stamps.data_ptr())
epoch = C.c_uint(0)
status = torch.zeros(4, dtype=torch.int32, device="cuda")
names = {0: "leaf start", 1: "G1 partial published", 2: "deferred update done (behind the G2 publish since round 5)", 3: "R1^-1 seen", 4: "q, G2 partial published", 5: "product done",
         6: "first hand-off (U'^-1) seen", 7: "V, (owner: correction), Z published", 12: "second hand-off (T) seen", 8: "all Z seen", 9: "fold, W published", 10: "all W seen",
         11: "next columns updated"}
order = [0, 1, 3, 4, 2, 5, 6, 7, 12, 8, 9, 10, 11]
fnames = {16: "all G1 seen", 17: "G1 summed", 18: "chol", 19: "R1^-1 published", 20: "all G2 seen", 21: "G2 summed", 22: "LU", 25: "first hand-off published (waves 1-3)",
          23: "U", 24: "T, R, published"}
forder = [16, 17, 18, 19, 20, 21, 22, 25, 23, 24]
for mk, wh in [(2048, 64), (4096, 256), (8192, 256), (4096, 128)]:
    P = torch.from_numpy(np.ascontiguousarray(np.random.default_rng(1).random((wh, mk)))).cuda()
    V = torch.zeros((wh, mk), dtype=torch.float64, device="cuda")
    T = torch.zeros((wh, wh), dtype=torch.float64, device="cuda")
    G = torch.zeros((wh, wh), dtype=torch.float64, device="cuda")
    tau = torch.zeros(wh, dtype=torch.float64, device="cuda")
    for rep in range(3):
        b = P.clone()
        torch.cuda.synchronize()
        assert f(None, b.data_ptr(), mk, mk, wh, tau.data_ptr(), T.data_ptr(), wh, V.data_ptr(), mk, None, wh, ws.data_ptr(),
                 C.byref(epoch), status.data_ptr()) == 0
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(8, 32)
    print(f"--- {mk} x {wh}: leaf 0 and leaf {wh // 32 - 1} (us since leaf start; delta)")
    for li in sorted({0, 1, wh // 32 - 1}):
        t0 = st[li][0]
        prev = t0
        line = []
        for k in order:
            t = st[li][k]
            line.append(f"{names[k]}: {(t - t0) / 100:.1f} (+{(t - prev) / 100:.1f})")
            prev = t
        print(f" leaf {li}: " + " | ".join(line))
        print('   factor wg: ' + ' | '.join(f"{fnames[k]}: {(st[li][k] - t0) / 100:.1f}" for k in forder))
    if wh > 32:
        print(f" leaf 0 start -> leaf 1 start: {(st[1][0] - st[0][0]) / 100:.1f} us")
