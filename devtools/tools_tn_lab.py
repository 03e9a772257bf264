"""isolated whole-chip rate of the wide TN product Wt = A2^T (V T) (qrd_gemm_tn_update) at update shapes:
   python devtools/tools_tn_lab.py 15872x256x16128 [more MxNxK]   (M = columns of A2, N = nb, K = panel height)"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time, json
import torch
import cuda_qr_amd as q

for spec in sys.argv[1:]:
    M, N, K = (int(x) for x in spec.split("x"))
    A = torch.rand((M, K), dtype=torch.float64, device="cuda")          # column-major K x M
    B = torch.rand((N, K), dtype=torch.float64, device="cuda")          # column-major K x N
    Cc = torch.empty((N, M), dtype=torch.float64, device="cuda")        # column-major M x N
    slabs = torch.empty(1 << 24, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    f = q.lib.qrd_gemm_tn_update
    def run():
        q.check(f(None, M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cc.data_ptr(), M, slabs.data_ptr(), 1 << 24))
    run(); q.check(q.lib.qrd_device_sync())
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        q.check(q.lib.qrd_device_sync())
        best = min(best, (time.perf_counter() - t0) / 10)
    print(json.dumps({"M": M, "N": N, "K": K, "ms": best * 1e3, "tflops": 2.0 * M * N * K / best / 1e12}), flush=True)
