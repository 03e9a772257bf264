"""stage checks of the transposition-free passes: G1 against A^T A, Q against A R1^-1, G2 against Q^T Q.
   MI355XQR_CQR_DIRECT=1|2|3 python devtools/tools_cqr_direct_debug.py mk [lda_pad]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C, sys
import numpy as np, torch
import cuda_qr_amd as qr
mk = int(sys.argv[1]); pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = 128; lda = mk + pad
L = qr.lib
qr.check(L.qrd_init(), "init")
L.qrd_panel_cqr_ws_doubles.restype = C.c_size_t
L.qrd_panel_cqr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
ws = torch.full((int(L.qrd_panel_cqr_ws_doubles()),), float("nan"), dtype=torch.float64, device="cuda")
status = torch.zeros(4, dtype=torch.int32, device="cuda")
torch.manual_seed(3)
A = torch.rand((w, lda), dtype=torch.float64, device="cuda") - 0.5
A0 = A.clone()
V = torch.zeros((w, lda), dtype=torch.float64, device="cuda"); T = torch.zeros((w, w), dtype=torch.float64, device="cuda"); tau = torch.zeros(w, dtype=torch.float64, device="cuda")
rc = L.qrd_panel_cqr(None, A.data_ptr(), lda, mk, w, tau.data_ptr(), T.data_ptr(), w, V.data_ptr(), lda, ws.data_ptr(), status.data_ptr())
torch.cuda.synchronize()
a = A0[:, :mk].cpu().numpy().T                 # mk x w
wsh = ws[: 13 * 128 * 128].cpu().numpy()
G1 = wsh[0:16384].reshape(128, 128)            # column-major, symmetric
G2 = wsh[16384:2 * 16384].reshape(128, 128)
R1I = wsh[3 * 16384:4 * 16384].reshape(128, 128)
ref1 = a.T @ a
print("rc", rc, "status", status.cpu().numpy())
e1 = np.abs(G1 - ref1) / np.abs(ref1).max()
print("G1 rel err max %.3e  (nan %d)" % (np.nanmax(e1), np.isnan(G1).sum()))
if np.nanmax(e1) > 1e-12 or np.isnan(G1).any():
    bad = np.argwhere(~(e1 < 1e-12)); print("  bad tiles (16x16):", sorted({(int(i) // 16, int(j) // 16) for i, j in bad})[:40])
# Q lives in Vw when the panel was refused; otherwise Vw holds V -- compare only when refused or check via G2
q_ref = a @ np.triu(R1I)
g2_ref = q_ref.T @ q_ref
e2 = np.abs(G2 - g2_ref)
print("G2 abs err max %.3e (nan %d), |G2 - I| %.3e" % (np.nanmax(e2), np.isnan(G2).sum(), np.nanmax(np.abs(G2 - np.eye(128)))))
if np.nanmax(e2) > 1e-12 or np.isnan(G2).any():
    bad = np.argwhere(~(e2 < 1e-12)); print("  bad tiles (16x16):", sorted({(int(i) // 16, int(j) // 16) for i, j in bad})[:40])
if status.cpu().numpy()[0]:
    q = V[:, :mk].cpu().numpy().T
    eq = np.abs(q - q_ref)
    print("Q (refused panel: still in Vw) abs err max %.3e" % np.nanmax(eq))
    if np.nanmax(eq) > 1e-12:
        bad = np.argwhere(~(eq < 1e-12)); print("  bad rows mod 16:", sorted({int(i) % 16 for i, j in bad}), "bad col tiles:", sorted({int(j) // 16 for i, j in bad}), "n bad", len(bad))
        i, j = bad[0]; print("  first bad", i, j, q[i, j], q_ref[i, j])
