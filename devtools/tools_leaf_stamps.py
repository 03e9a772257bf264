"""Phase times inside cholq2_kernel / hr3_kernel (development build `make -C cuda-qr_amd stamps`):
tools_leaf_stamps.py [rows ...]   -- one CholeskyQR2 leaf of 32 columns per height, stamps of workgroup 0 / thread 0."""
import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(here, "cuda-qr_amd", "libmi355xqr_stamps.so"))
vp = C.c_void_p
lib.qrd_panel_ws_size.restype = C.c_size_t; lib.qrd_panel_ws_size.argtypes = [C.c_int]
lib.qrd_panel_cholqr.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.c_size_t, C.c_int]
lib.qrd_dbg_read_stamps.argtypes = [C.POINTER(C.c_longlong)]
assert lib.qrd_init() == 0
names = {0: "cholq2 start", 1: "  G1 slabs summed", 2: "  Cholesky done", 3: "  q = a R1^-1 done", 4: "  V stores issued",
         5: "  Gram MFMA done", 6: "cholq2 end", 16: "hr3 start", 17: "  G2 slabs summed + guard", 18: "  Cholesky R2 done",
         19: "  LU done", 20: "  U, L^-1, R written", 21: "hr3 end (T)"}
for mk in [int(x) for x in sys.argv[1:]] or [8192, 2048]:
    w = 32
    P = torch.rand((w, mk), dtype=torch.float64, device="cuda")
    tau = torch.zeros(w, dtype=torch.float64, device="cuda"); T = torch.zeros((w, w), dtype=torch.float64, device="cuda")
    V = torch.zeros((w, mk), dtype=torch.float64, device="cuda")
    ws = torch.zeros(int(lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    cws = torch.zeros(4 * 32 * 32 + 16, dtype=torch.float64, device="cuda")
    slabs = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    for rep in range(3):
        P.uniform_()
        torch.cuda.synchronize()
        rc = lib.qrd_panel_cholqr(None, P.data_ptr(), mk, mk, w, tau.data_ptr(), T.data_ptr(), w, V.data_ptr(), mk, ws.data_ptr(), mk,
                                  cws.data_ptr(), slabs.data_ptr(), 1 << 20, 0)
        assert rc == 0, rc
        torch.cuda.synchronize()
    st = (C.c_longlong * 64)()
    assert lib.qrd_dbg_read_stamps(st) == 0
    print("leaf %d x 32   (100 MHz wall clock, us since the kernel's first stamp)" % mk)
    for base, ks in ((0, range(0, 7)), (16, range(16, 22))):
        prev = st[base]
        for k in ks:
            print("  %-28s %7.2f  (+%.2f)" % (names[k], (st[k] - st[base]) / 100.0, (st[k] - prev) / 100.0))
            prev = st[k]
    print("  hr3 start - cholq2 end: %.2f us" % ((st[16] - st[6]) / 100.0))
