"""Reference point only: what does the vendor DGEMM (rocBLAS/hipBLASLt through torch) reach on the update's shapes?"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import time, json, torch
for (M, N, K) in ((16128, 15872, 256), (8192, 7936, 256), (16128, 15872, 4096), (256, 15872, 16128)):
    A = torch.rand((M, K), dtype=torch.float64, device="cuda"); B = torch.rand((K, N), dtype=torch.float64, device="cuda")
    C = torch.rand((M, N), dtype=torch.float64, device="cuda")
    for variant in ("rowmajor", "colmajor"):
        if variant == "colmajor":            # same memory seen as column-major operands: C^T = B^T A^T
            a, b, c = B.t().contiguous().t(), A.t().contiguous().t(), C
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                if variant == "rowmajor": C.addmm_(A, B, alpha=-1.0)
                else: C.addmm_(b, a, alpha=-1.0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"vendor_dgemm": [M, N, K], "variant": variant, "ms": round(dt * 1e3, 3), "tflops": round(2.0 * M * N * K / dt / 1e12, 2)}), flush=True)
    del A, B, C
