"""which library call in the process slows the two-stream schedule afterwards?  python devtools/tools_seq_probe2.py"""
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
if len(_sys.argv) > 1:
    _os.environ["MI355XQR_LOOKAHEAD"] = "1"; _os.environ["MI355XQR_PANEL_CUS"] = "64"
    keep = q.Plan(256, 64, 32, 32)          # a tiny plan with its CU-masked stream pair, kept alive for the whole process
    del _os.environ["MI355XQR_LOOKAHEAD"]; del _os.environ["MI355XQR_PANEL_CUS"]
    print("(keeping a small plan with masked streams alive)")
print("fresh 8192^2                      %.2f ms" % t(8192, 8192, 256))
a = torch.rand((8192, 8192), dtype=torch.float64, device="cuda"); b = a @ a; torch.cuda.synchronize(); del b
print("after torch f64 matmul 8192       %.2f ms" % t(8192, 8192, 256))
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    b = a @ a
torch.cuda.synchronize(); del b
print("after matmul on a side stream     %.2f ms" % t(8192, 8192, 256))
x = torch.rand((8192, 8192), dtype=torch.float64, device="cuda"); qq, tt = torch.geqrf(x.mT); torch.cuda.synchronize(); del qq, tt
print("after torch.geqrf 8192            %.2f ms" % t(8192, 8192, 256))
x = torch.rand((16384, 16384), dtype=torch.float64, device="cuda"); qq, tt = torch.geqrf(x.mT); torch.cuda.synchronize(); del qq, tt, x
print("after torch.geqrf 16384           %.2f ms" % t(8192, 8192, 256))
