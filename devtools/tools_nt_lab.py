"""isolated whole-chip rate of the trailing update A2 -= V Wt^T (qrd_gemm_nt) at update shapes:
   python devtools/tools_nt_lab.py 16384x16128x256 [more MxNxK]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time, json, ctypes as C
import torch
import cuda_qr_amd as q

f = q.lib.qrd_gemm_nt
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
for spec in sys.argv[1:]:
    M, N, K = (int(x) for x in spec.split("x"))
    A = torch.rand((K, M), dtype=torch.float64, device="cuda")          # column-major M x K
    Bt = torch.rand((K, N), dtype=torch.float64, device="cuda")         # column-major N x K
    Cc = torch.rand((N, M), dtype=torch.float64, device="cuda")         # column-major M x N
    torch.cuda.synchronize()
    def run():
        q.check(f(None, M, N, K, -1, A.data_ptr(), M, Bt.data_ptr(), N, Cc.data_ptr(), M, -1, None))
    run(); q.check(q.lib.qrd_device_sync())
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        q.check(q.lib.qrd_device_sync())
        best = min(best, (time.perf_counter() - t0) / 10)
    print(json.dumps({"M": M, "N": N, "K": K, "ms": best * 1e3, "tflops": 2.0 * M * N * K / best / 1e12}), flush=True)
