"""Which call leaves later factorisations slower?  (tools_perf.py with CHECK=1 showed shapes late in a list up to 2x slower.)
Times geqrf of a fresh 6000^2 plan after each candidate step.  python devtools/tools_slowdown_probe.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import time, torch
import cuda_qr_amd as q

def t_geqrf(m=6000, n=6000, reps=3):
    p = q.Plan(m, n, 0, 0)
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    best = 1e9
    for r in range(reps):
        p.fill_uniform(dA, m, m, n, seed=12); p.sync()
        t0 = time.perf_counter(); p.geqrf(dA, m, n, m, dtau); p.sync(); best = min(best, time.perf_counter() - t0)
    p.close()
    return best * 1e3

def z(r, c): return torch.zeros((c, r), dtype=torch.float64, device="cuda")

print("fresh process            %.2f ms" % t_geqrf()); print("again                    %.2f ms" % t_geqrf())
m = n = 4100
p = q.Plan(m, n, 256, 32); dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
p.fill_uniform(dA, m, m, n, seed=12); p.geqrf(dA, m, n, m, dtau); p.sync()
print("after a 4100^2 geqrf     %.2f ms" % t_geqrf())
dR = z(n, n); torch.cuda.synchronize(); p.extract_r(dA, m, n, m, dR, n, n); p.sync()
print("after extract_r          %.2f ms" % t_geqrf())
dQ = z(m, n); torch.cuda.synchronize(); p.applyq(dA, m, n, m, dtau, dQ, n, m, True); p.sync()
print("after applyq             %.2f ms" % t_geqrf())
dQR = z(m, n); torch.cuda.synchronize(); p.gemm("N", m, n, n, 1.0, dQ, m, dR, n, 0.0, dQR, m); p.sync()
print("after gemm               %.2f ms" % t_geqrf())
d, a = p.diffnorm(dQR, m, m, n, seed=12)
print("after diffnorm           %.2f ms" % t_geqrf())
p.close()
print("after closing that plan  %.2f ms" % t_geqrf())
del dQ, dR, dQR, dA; torch.cuda.empty_cache()
print("after freeing its buffers %.2f ms" % t_geqrf())
