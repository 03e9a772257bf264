import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import sys, torch
import cuda_qr_amd as q
mode = sys.argv[1]
if mode == "copy":
    q.probe_copy_gbps()
elif mode == "plan":
    p = q.Plan(1024, 512, 128, 32); p.sync()
elif mode == "fill":
    p = q.Plan(1024, 512, 128, 32)
    A = torch.empty((512, 1024), dtype=torch.float64, device="cuda")
    p.fill_uniform(A, 1024, 1024, 512); p.sync()
elif mode == "geqrf":
    p = q.Plan(1024, 512, 128, 32)
    A = torch.empty((512, 1024), dtype=torch.float64, device="cuda"); t = torch.empty(512, dtype=torch.float64, device="cuda")
    p.fill_uniform(A, 1024, 1024, 512); p.geqrf(A, 1024, 512, 1024, t); p.sync()
print("ok", mode)
