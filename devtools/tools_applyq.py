"""time of forming the thin / full Q (qr_applyq_dev, the explicitQR path) next to the factorisation"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import time, json, torch
import cuda_qr_amd as q
for (m, n, nb) in ((16384, 16384, 256), (8192, 8192, 256), (4096, 4096, 128), (262144, 512, 128)):
    p = q.Plan(m, n, nb, 32)
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    dQ = torch.empty((n, m), dtype=torch.float64, device="cuda")
    p.fill_uniform(dA, m, m, n, seed=12); p.sync()
    p.geqrf(dA, m, n, m, dtau); p.sync()
    best = 1e30
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p.applyq(dA, m, n, m, dtau, dQ, n, m, True); p.sync()
        best = min(best, time.perf_counter() - t0)
    fl = 2.0 * m * n * n - 2.0 * n ** 3 / 3.0
    print(json.dumps({"m": m, "n": n, "nb": nb, "form_thin_q_ms": round(best * 1e3, 2), "tflops": round(fl / best / 1e12, 2)}), flush=True)
    p.close()
