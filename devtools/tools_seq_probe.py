"""does what ran before in the process change a factorisation's time?  python devtools/tools_seq_probe.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import time, torch
import cuda_qr_amd as q
def t(m, n, nb, reps=3):
    p = q.Plan(m, n, nb, 32)
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    best = 1e30
    for r in range(reps + 1):
        p.fill_uniform(dA, m, m, n, seed=12); p.sync()
        t0 = time.perf_counter(); p.geqrf(dA, m, n, m, dtau); p.sync(); dt = time.perf_counter() - t0
        if r: best = min(best, dt)
    p.close(); del dA, dtau
    return best * 1e3
print("fresh 8192^2            %.2f ms" % t(8192, 8192, 256))
print("again                   %.2f ms" % t(8192, 8192, 256))
print("16384^2                 %.2f ms" % t(16384, 16384, 256))
print("8192^2 after 16384^2    %.2f ms" % t(8192, 8192, 256))
torch.cuda.empty_cache()
print("8192^2 after empty_cache %.2f ms" % t(8192, 8192, 256))
buf = torch.rand((4096, 4096), dtype=torch.float64, device="cuda"); a, tau = torch.geqrf(buf); torch.cuda.synchronize(); del a, tau, buf
print("8192^2 after torch.geqrf %.2f ms" % t(8192, 8192, 256))
print("4096^2                  %.2f ms" % t(4096, 4096, 256))
buf = torch.rand((16384, 16384), dtype=torch.float64, device="cuda"); a, tau = torch.geqrf(buf.mT); torch.cuda.synchronize(); del a, tau, buf
print("8192^2 after torch.geqrf 16384^2 %.2f ms" % t(8192, 8192, 256))
torch.cuda.empty_cache()
print("8192^2 after empty_cache         %.2f ms" % t(8192, 8192, 256))
print("16384^2                          %.2f ms" % t(16384, 16384, 256))
print("6144^2                           %.2f ms" % t(6144, 6144, 256))
