"""stage-by-stage check of the full-width tall panel (qr_panel_cqr.hip) against numpy: python devtools/tools_cqr_debug.py mk w"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import ctypes as C, sys
import numpy as np, torch
import cuda_qr_amd as qr
from gpu_util import dev, host, zeros
mk, w = int(sys.argv[1]), int(sys.argv[2])
L = qr.lib if len(sys.argv) <= 3 else C.CDLL(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'cuda-qr_amd', 'libmi355xqr_cqstamps.so'))
qr.check(L.qrd_init(), "init")
L.qrd_panel_cqr_ws_doubles.restype = C.c_size_t
for f in (L.qrd_panel_cqr_g1, L.qrd_panel_cqr_g2): f.restype = C.c_void_p; f.argtypes = [C.c_void_p]
L.qrd_device_sync.restype = C.c_int
L.qrd_panel_cqr_stage1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
L.qrd_panel_cqr_stage2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
L.qrd_gemm_tn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
P = np.random.default_rng(mk + w).random((mk, w))
dA, dV = dev(P), zeros(mk, w)
dT, dtau = zeros(w, w), zeros(w, 1)
ws = torch.zeros(int(L.qrd_panel_cqr_ws_doubles()), dtype=torch.float64, device="cuda")
status = torch.zeros(4, dtype=torch.int32, device="cuda")
cap = 1 << 22
slabs = torch.zeros(cap, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def mat(idx, colmajor=False):
    torch.cuda.synchronize()
    a = ws.cpu().numpy()[idx * 128 * 128:(idx + 1) * 128 * 128].reshape(128, 128)
    return (a.T if colmajor else a)[:w, :w].copy()
def err(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
def mixed(M):
    """what the streaming passes multiply by since the end of round 5: the upper-triangular factor's off-diagonal 32 x 32 blocks and the
    INVERSES of its diagonal blocks (block back substitution instead of a product with the full inverse)"""
    X = np.triu(M).copy()
    for o in range(0, M.shape[0], 32):
        X[o:o + 32, o:o + 32] = np.linalg.inv(np.triu(M[o:o + 32, o:o + 32]))
    return X
g1, g2 = L.qrd_panel_cqr_g1(ws.data_ptr()), L.qrd_panel_cqr_g2(ws.data_ptr())
print("tn", L.qrd_gemm_tn(None, w, w, mk, 1.0, dA.data_ptr(), mk, dA.data_ptr(), mk, 0.0, g1, 128, slabs.data_ptr(), cap, None, 0))
G1 = P.T @ P
print("G1 err", err(np.triu(mat(0, True)), np.triu(G1)))
print("stage1", L.qrd_panel_cqr_stage1(None, dA.data_ptr(), mk, mk, w, dV.data_ptr(), mk, ws.data_ptr(), status.data_ptr()))
L.qrd_device_sync()
print("status", status.cpu().numpy())
R1 = np.linalg.cholesky(G1).T
print("R1 err", err(mat(2), R1), " R1 blocks / diagonal-block inverses err", err(mat(3), mixed(R1)))
Q = host(dV)
Qref = P @ np.linalg.inv(mat(2))
print("Q err", err(Q, Qref), " |QtQ - I|", np.abs(Q.T @ Q - np.eye(w)).max())
print("tn", L.qrd_gemm_tn(None, w, w, mk, 1.0, dV.data_ptr(), mk, dV.data_ptr(), mk, 0.0, g2, 128, slabs.data_ptr(), cap, None, 0))
G2 = Q.T @ Q
print("G2 err", err(np.triu(mat(1, True)), np.triu(G2)))
print("stage2", L.qrd_panel_cqr_stage2(None, dA.data_ptr(), mk, mk, w, dtau.data_ptr(), dT.data_ptr(), w, dV.data_ptr(), mk, ws.data_ptr(), status.data_ptr()))
L.qrd_device_sync()
print("status", status.cpu().numpy())
R2 = np.linalg.cholesky(G2).T
print("R2 err", err(mat(5), R2))
LU, S = mat(6), ws.cpu().numpy()[12 * 128 * 128:12 * 128 * 128 + w]
L1, Up = np.tril(LU, -1) + np.eye(w), np.triu(LU)
print("LU err", err(L1 @ Up, Q[:w] - S[:, None] * mat(5)))
print("U' blocks / diagonal-block inverses err", err(mat(4), mixed(Up)))
print("R err", err(np.triu(mat(9)), S[:, None] * mat(5) @ mat(2)))
U = Up @ np.linalg.inv(mat(5))
print("T err", err(np.triu(mat(10)), -U @ np.diag(S) @ np.linalg.inv(L1).T))
V = host(dV)
out = host(dA)
print("V err", err(V[w:], (Q[w:]) @ np.linalg.inv(Up)), " top", err(V[:w], L1))
T = host(dT)
QtP = P - V @ (T.T @ (V.T @ P))
print("below-diagonal of Q^T P", np.abs(np.tril(QtP, -1)).max(), " R match", err(np.triu(QtP[:w]), np.triu(out[:w])))
if len(sys.argv) > 3:
    st = ws.cpu().numpy()[12 * 128 * 128 + 128:12 * 128 * 128 + 128 + 64].view(np.uint64)
    names = {1: "chol", 2: "R1 out", 4: "blocks + diagonal inverses out", 9: "G2 -> R2, R2^-1", 10: "W load + LU", 11: "LU out"}
    t = lambda i: int(st[i]) * 0.01
    print('  LU kernel (us): diagonal block 0 %.1f | S R2 + U12 + L21 (all waves) %.1f | wave 0: next block update %.1f' % (t(25) - t(24), t(26) - t(25), t(27) - t(26)))
    print('  LU kernel start (us): Q_top requested + G2 into LDS %.1f | distance from I, first-order factor %.1f | R2 and R2^-1 stores issued %.1f | to the barrier %.1f'
          % (t(28) - t(8), t(29) - t(28), t(30) - t(29), t(9) - t(30)))
    for i, prev in ((1, 0), (2, 1), (4, 2), (9, 8), (10, 9), (11, 10)):     # (stamp 3 went with the inverse levels of the Cholesky kernel)
        print("  %-32s %8.1f us" % (names[i], (int(st[i]) - int(st[prev])) * 0.01))
