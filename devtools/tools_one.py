import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import sys, torch
import cuda_qr_amd as q
m, n, nb = (int(x) for x in sys.argv[1].split("x"))
p = q.Plan(m, n, nb, 32)
dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
for r in range(2):
    p.fill_uniform(dA, m, m, n, seed=12); p.sync()
    p.geqrf(dA, m, n, m, dtau); p.sync()
p.close()
