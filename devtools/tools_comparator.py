"""Comparator only (the role MAGMA plays in the reference, qr.cu:555-565): rocSOLVER dgeqrf through torch.geqrf next to
this library on the same shapes, matrix resident in HBM.  Not part of the product path or of bench.py."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import json, time, sys
import torch
import cuda_qr_amd as q

def flops(m, n): return 2.0 * m * n * n - 2.0 * n ** 3 / 3.0

def time_ours(m, n, nb, reps=3):
    p = q.Plan(m, n, nb, 32)
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    best = 1e30
    for r in range(reps + 1):
        p.fill_uniform(dA, m, m, n, seed=12); p.sync()
        t0 = time.perf_counter(); p.geqrf(dA, m, n, m, dtau); p.sync(); dt = time.perf_counter() - t0
        if r: best = min(best, dt)
    p.close()
    return best

def time_rocsolver(m, n, reps=3):
    # torch stores row-major; geqrf wants column-major: give it a tensor whose .mT is contiguous so that no copy is timed
    p = q.Plan(64, 32)
    buf = torch.empty((n, m), dtype=torch.float64, device="cuda")
    best = 1e30
    for r in range(reps + 1):
        p.fill_uniform(buf, m, m, n, seed=12); p.sync()
        A = buf.mT                      # (m, n) view, column-major storage
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a, tau = torch.geqrf(A)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if r: best = min(best, dt)
        del a, tau
    p.close()
    return best

for (m, n, nb) in ((16384, 16384, 256), (8192, 8192, 256), (4096, 4096, 128), (262144, 512, 128), (65536, 256, 128)):
    t1 = time_ours(m, n, nb)
    try:
        t2 = time_rocsolver(m, n)
    except Exception as e:
        t2 = None; print("rocsolver failed:", repr(e)[:200], file=sys.stderr)
    print(json.dumps({"shape": [m, n], "mi355xqr_ms": round(t1 * 1e3, 2), "mi355xqr_gflops": round(flops(m, n) / t1 / 1e9, 1),
                      "rocsolver_dgeqrf_ms": round(t2 * 1e3, 2) if t2 else None,
                      "rocsolver_gflops": round(flops(m, n) / t2 / 1e9, 1) if t2 else None,
                      "note": "torch.geqrf (hipSOLVER/rocSOLVER); includes its output allocation/copy of A"}), flush=True)
