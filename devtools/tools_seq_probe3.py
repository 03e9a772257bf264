"""does a plan created BEFORE another library spins up its stream pool keep its speed?  python devtools/tools_seq_probe3.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import time, torch
import cuda_qr_amd as q
m = n = 8192
def run(p, dA, dtau, reps=3):
    best = 1e30
    for r in range(reps + 1):
        p.fill_uniform(dA, m, m, n, seed=12); p.sync()
        t0 = time.perf_counter(); p.geqrf(dA, m, n, m, dtau); p.sync(); dt = time.perf_counter() - t0
        if r: best = min(best, dt)
    return best * 1e3
dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
p_old = q.Plan(m, n, 256, 32)
print("plan A, fresh process                         %.2f ms" % run(p_old, dA, dtau))
a = torch.rand((4096, 4096), dtype=torch.float64, device="cuda")
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    b = a @ a
torch.cuda.synchronize(); del b
print("plan A (created before) after side stream     %.2f ms" % run(p_old, dA, dtau))
p_new = q.Plan(m, n, 256, 32)
print("plan B (created after the side stream)        %.2f ms" % run(p_new, dA, dtau))
print("plan A again                                  %.2f ms" % run(p_old, dA, dtau))
del s2
torch.cuda.synchronize()
p_new2 = q.Plan(m, n, 256, 32)
print("plan C (after dropping the torch stream obj)  %.2f ms" % run(p_new2, dA, dtau))
