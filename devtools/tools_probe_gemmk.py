"""GEMM NN rate against K (is the 26% gap to the MFMA peak in the main loop or in the per-tile prologue/epilogue?)"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import ctypes as C, time, json
import torch
import cuda_qr_amd as q
lib = q.lib
M, N = 16128, 15872
p = q.Plan(1024, 1024)
st = p.stream
for K in (64, 128, 256, 512, 1024, 4096):
    A = torch.rand((K, M), dtype=torch.float64, device="cuda"); B = torch.rand((N, K), dtype=torch.float64, device="cuda")
    Cm = torch.rand((N, M), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for beta in (1.0, 0.0):
        for rep in range(2):
            p.sync(); t0 = time.perf_counter()
            for _ in range(3):
                q.check(lib.qrd_gemm_nn(st, M, N, K, -1.0, A.data_ptr(), M, B.data_ptr(), K, beta, Cm.data_ptr(), M))
            p.sync(); dt = (time.perf_counter() - t0) / 3
        print(json.dumps({"gemm_nn": [M, N, K], "beta": beta, "ms": round(dt * 1e3, 3), "tflops": round(2.0 * M * N * K / dt / 1e12, 2)}), flush=True)
    del A, B, Cm
