"""tall leaf through the streaming Cholesky-Q pass with inputs scaled by 1e-150 / 1e+150: R must scale exactly like the input"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np, torch
import cuda_qr_amd as q
from oracle import oracle as O
m, n = 65536, 64
rng = np.random.default_rng(3)
A = rng.standard_normal((m, n))
def R_of(A):
    p = q.Plan(m, n, 32, 32)
    dA = torch.from_numpy(np.ascontiguousarray(A.T)).cuda(); dtau = torch.zeros(n, dtype=torch.float64, device="cuda")
    dR = torch.zeros((n, n), dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
    p.geqrf(dA, m, n, m, dtau); p.extract_r(dA, m, n, m, dR, n, n); p.sync()
    R = np.asfortranarray(dR.cpu().numpy().T); p.close()
    return O.sign_normalise(R)
R0 = R_of(A)
Rl = O.sign_normalise(np.linalg.qr(A, mode="r"))
print("unscaled vs LAPACK %.2e" % (np.linalg.norm(R0 - Rl) / np.linalg.norm(Rl)))
for sc in (1e-150, 1e150, 2.0 ** -500, 3e-160):
    Rs = R_of(A * sc)
    print("scale %.3g: |R_s/scale - R_0| / |R_0| = %.2e  finite %s" % (sc, np.linalg.norm(Rs / sc - R0) / np.linalg.norm(R0), np.isfinite(Rs).all()))
