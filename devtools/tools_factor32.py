"""time of the 32 x 32 small-factor routines, one wave, LDS -> LDS (profiles/r05_leaf_phase_stamps.txt): python devtools/tools_factor32.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import cuda_qr_amd as qr
L = qr.lib
qr.check(L.qrd_init(), "init")
L.qrd_dbg_chol32.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3
L.qrd_dbg_lu32.argtypes = [C.c_void_p] * 8 + [C.c_int] * 3
rng = np.random.default_rng(0)
for nmat in (1, 256):
    A = rng.standard_normal((nmat, 200, 32)); G = np.einsum('qki,qkj->qij', A, A)
    dG = torch.from_numpy(G).cuda(); dR = torch.zeros_like(dG); dX = torch.zeros_like(dG)
    ok = torch.zeros(nmat, dtype=torch.int32, device="cuda"); tk = torch.zeros(nmat, dtype=torch.int64, device="cuda")
    W = np.stack([np.linalg.qr(rng.standard_normal((300, 32)))[0][:32] for _ in range(nmat)]); R2 = np.stack([np.eye(32)] * nmat)
    dW, dR2 = torch.from_numpy(W).cuda(), torch.from_numpy(R2).cuda()
    dLU, dLi, dUit = torch.zeros_like(dW), torch.zeros_like(dW), torch.zeros_like(dW); dS = torch.zeros((nmat, 32), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    reps = 200
    for name, variant in (("register recurrence (CholAugStep)", 0), ("matrix cores (chol32_mfma)", 1)):
        for _ in range(2):
            L.qrd_dbg_chol32(None, dG.data_ptr(), dR.data_ptr(), dX.data_ptr(), ok.data_ptr(), tk.data_ptr(), nmat, variant, reps); torch.cuda.synchronize()
        t = tk.cpu().numpy() / reps * 10.0     # 100 MHz ticks -> ns
        print("chol 32x32 + inverse  %-40s %4d wave(s) at once: %.2f us per factorisation (min %.2f max %.2f)" % (name, nmat, t.mean() / 1e3, t.min() / 1e3, t.max() / 1e3))
    for name, variant in (("register recurrences (Hr3Lu, then UpperInv)", 0), ("matrix cores (lu32_mfma, both inverses)", 1)):
        for _ in range(2):
            L.qrd_dbg_lu32(None, dW.data_ptr(), dR2.data_ptr(), dLU.data_ptr(), dS.data_ptr(), dLi.data_ptr(), dUit.data_ptr(), tk.data_ptr(), nmat, variant, reps); torch.cuda.synchronize()
        t = tk.cpu().numpy() / reps * 10.0
        print("modified LU + inverses %-40s %4d wave(s) at once: %.2f us per factorisation (min %.2f max %.2f)" % (name, nmat, t.mean() / 1e3, t.min() / 1e3, t.max() / 1e3))
