"""single process, one GPU: does the stacked QR of step i really run under the local QR of step i+1?"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import time, json, torch
import cuda_qr_amd as q
from cuda_qr_amd import tsqr as T
m, n, P = 262144, 512, 8
be = T.HipBackend(q, m, n, P, 128, 32)
A = [be.new_matrix(m, n) for _ in range(3)]
for i, a in enumerate(A): be.fill(a, m, n, 0, m, 12 + i)
Rl = be.new_matrix(n, n); R = be.new_matrix(n, n); S = be.new_matrix(P * n, n)
be.plan.fill_uniform(S, P * n, P * n, n, seed=5); be.plan.sync()
for mode in ("sequential", "pipelined"):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(6):
            be.local_factor(A[i % 3], Rl)
            if mode == "pipelined":
                be.stack_sync(); be.stack_factor(S, R, wait=False)
            else:
                be.stack_factor(S, R)
        be.stack_sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(json.dumps({"mode": mode, "ms_per_step": round(dt * 1e3, 3)}), flush=True)
be.close()
