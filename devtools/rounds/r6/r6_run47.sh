#!/bin/bash
# round 6, run 47: explicitQR (host pointers, dense m x m Q) and qr_applyq_dev on odd heights against their even neighbours
cd $GRAFT_REPO_ROOT
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, numpy as np, torch
import cuda_qr_amd as qr
for (m, n) in [(2048, 700), (2049, 700), (4096, 1024), (4097, 1024), (5000, 2000), (5001, 2000)]:
    A = np.random.default_rng(1).random((m, n))
    F, tau = qr.mmqr(A)
    qr.explicit_qr(F, tau)
    t0 = time.perf_counter(); Q, R = qr.explicit_qr(F, tau); t1 = time.perf_counter()
    print("explicitQR host pointers %5d x %4d: %.1f ms" % (m, n, (t1 - t0) * 1e3), flush=True)
    p = qr.Plan(m, n, 0, 0)
    dA = torch.from_numpy(np.ascontiguousarray(F.T)).cuda(); dtau = torch.from_numpy(tau[:n].copy()).cuda(); dQ = torch.zeros((n, m), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for r in range(3):
        t0 = time.perf_counter(); p.applyq(dA, m, n, m, dtau, dQ, n, m, True); p.sync(); t1 = time.perf_counter()
    print("   qr_applyq_dev thin Q             : %.2f ms" % ((t1 - t0) * 1e3), flush=True)
    p.close()
PY
