"""time of one full-width tall panel (qrd_panel_cqr) by leading dimension: python devtools/tools_cqr_perf.py mk w [pad ...]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C, sys, time
import numpy as np, torch
import cuda_qr_amd as qr
mk, w = int(sys.argv[1]), int(sys.argv[2])
pads = [int(x) for x in sys.argv[3:]] or [0]
L = qr.lib if not _os.environ.get('CQR_LIB') else C.CDLL(_os.path.abspath(_os.environ['CQR_LIB']))
L.qrd_device_sync.restype = C.c_int
qr.check(L.qrd_init(), "init")
L.qrd_panel_cqr_ws_doubles.restype = C.c_size_t
L.qrd_panel_cqr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
ws = torch.zeros(int(L.qrd_panel_cqr_ws_doubles()), dtype=torch.float64, device="cuda")
status = torch.zeros(4, dtype=torch.int32, device="cuda")
for pad in pads:
    lda = mk + pad
    A0 = torch.rand((w, lda), dtype=torch.float64, device="cuda")
    A = A0.clone(); V = torch.zeros((w, lda), dtype=torch.float64, device="cuda")
    T = torch.zeros((w, w), dtype=torch.float64, device="cuda"); tau = torch.zeros(w, dtype=torch.float64, device="cuda")
    best = 1e30
    for rep in range(5):
        A.copy_(A0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.qrd_panel_cqr(None, A.data_ptr(), lda, mk, w, tau.data_ptr(), T.data_ptr(), w, V.data_ptr(), lda, ws.data_ptr(), status.data_ptr())
        L.qrd_device_sync()
        dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    print("mk %d w %d lda %d: %.1f us  (rc %d status %s)" % (mk, w, lda, best * 1e6, rc, status.cpu().numpy()[:1]), flush=True)
