import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, numpy as np, torch
import cuda_qr_amd as q
from oracle import oracle as O
m, n, nb, grade = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
rng = np.random.default_rng(5)
A = rng.random((m, n)) * np.logspace(0, -grade, n)[None, :]
p = q.Plan(m, n, nb, 32)
dA = torch.from_numpy(np.ascontiguousarray(A.T)).cuda(); dtau = torch.zeros(n, dtype=torch.float64, device="cuda")
dR = torch.zeros((n, n), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
p.geqrf(dA, m, n, m, dtau); p.extract_r(dA, m, n, m, dR, n, n); p.sync()
R = np.asfortranarray(dR.cpu().numpy().T)
F = np.asfortranarray(dA.cpu().numpy().T); tau = dtau.cpu().numpy()
Rln = O.sign_normalise(np.linalg.qr(A, mode="r")); Rn = O.sign_normalise(R)
cn = np.linalg.norm(Rln, axis=0)
err = np.linalg.norm((Rn - Rln) / cn[None, :]) / np.sqrt(n)
dQ = torch.zeros((n, m), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()              # torch's fill runs on torch's stream, the plan's work on the plan's: order them
p.applyq(dA, m, n, m, dtau, dQ, n, m, True); p.sync()
Q = np.asfortranarray(dQ.cpu().numpy().T)
# independent Q from (V, tau) on the host: LAPACK dorgqr
from scipy.linalg import lapack
Qh, _, info = lapack.dorgqr(np.asfortranarray(F[:, :n]), tau[:n])
print("%d x %d nb %d grade 1e-%g: |dR| %.2e  device Q: resid %.2e orth %.2e | host Q from (V,tau): resid %.2e orth %.2e  tau range %.3g..%.3g nan %d" % (
    m, n, nb, grade, err, np.linalg.norm(A - Q @ R) / np.linalg.norm(A), np.linalg.norm(Q.T @ Q - np.eye(n)),
    np.linalg.norm(A - Qh @ R) / np.linalg.norm(A), np.linalg.norm(Qh.T @ Qh - np.eye(n)), tau.min(), tau.max(), int(np.isnan(F).sum())))

p.close()
