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
names = {0: "leaf start", 1: "image", 2: "gram1 waves", 3: "G1 published", 4: "G1 all seen", 20: "G1 summed(own)", 5: "G1 summed", 6: "chol1",
         7: "Q,image,Qtop", 8: "gram2 waves", 9: "G2 published", 10: "G2 all seen", 11: "G2 summed", 12: "LU | product", 13: "Z pub | tri",
         14: "T,out | V", 15: "Z all seen", 21: "fold(own)", 16: "W published", 17: "W all seen", 18: "update(next cols)", 6: "deferred upd | chol1"}
order = [0, 1, 2, 3, 4, 20, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 21, 16, 17, 18]
for mk, wh in [(4096, 32), (8192, 256), (4096, 64), (1024, 256)]:
    P = torch.from_numpy(np.ascontiguousarray(np.random.default_rng(1).random((wh, mk)))).cuda()
    V = torch.zeros((wh, mk), dtype=torch.float64, device="cuda")
    T = torch.zeros((wh, wh), dtype=torch.float64, device="cuda")
    G = torch.zeros((wh, wh), dtype=torch.float64, device="cuda")
    tau = torch.zeros(wh, dtype=torch.float64, device="cuda")
    for rep in range(3):
        b = P.clone()
        torch.cuda.synchronize()
        assert f(None, b.data_ptr(), mk, mk, wh, tau.data_ptr(), T.data_ptr(), wh, V.data_ptr(), mk, G.data_ptr(), wh, ws.data_ptr(),
                 C.byref(epoch), status.data_ptr()) == 0
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(8, 32)
    print(f"--- {mk} x {wh}: leaf 0 and leaf {wh // 32 - 1} (us since leaf start; delta)")
    for li in sorted({0, wh // 32 - 1}):
        t0 = st[li][0]
        prev = t0
        line = []
        for k in order:
            t = st[li][k]
            line.append(f"{names[k]}: {(t - t0) / 100:.1f} (+{(t - prev) / 100:.1f})")
            prev = t
        print(f" leaf {li}: " + " | ".join(line))
        print('   service wave 0 (us since leaf start): LU start %.1f, LU done %.1f, U/tri done %.1f, correction done %.1f' % tuple((st[li][k] - t0) / 100 for k in (23, 24, 25, 26)))
    if wh > 32:
        print(f" leaf 0 start -> leaf 1 start: {(st[1][0] - st[0][0]) / 100:.1f} us")
