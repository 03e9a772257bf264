"""Leaf-panel workload for the PMC / kernel-trace passes: the CholeskyQR2 + Householder-reconstruction leaf (and, with
argv[1] = tsqr, the Householder TSQR leaf) on tall mk x 32 panels, launched through the library's own leaf entry points."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import sys, json, ctypes as C
import torch
import cuda_qr_amd as q
lib = q.lib
mode = sys.argv[1] if len(sys.argv) > 1 else "cholqr"
p = q.Plan(1024, 1024)
st = p.stream
shapes = [(262144, 32), (65536, 32), (16384, 32)]
out = []
for (mk, w) in shapes:
    P0 = torch.rand((w, mk), dtype=torch.float64, device="cuda")
    tau = torch.zeros(w, dtype=torch.float64, device="cuda"); Tm = torch.zeros((w, w), dtype=torch.float64, device="cuda")
    V = torch.zeros((w, mk), dtype=torch.float64, device="cuda")
    ws = torch.zeros(int(lib.qrd_panel_ws_size(mk)), dtype=torch.float64, device="cuda")
    cws = torch.zeros(4 * 32 * 32 + 16, dtype=torch.float64, device="cuda"); slabs = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        Pm = P0.clone(); torch.cuda.synchronize()
        if mode == "tsqr":
            q.check(lib.qrd_panel_tsqr(st, Pm.data_ptr(), mk, mk, w, tau.data_ptr(), Tm.data_ptr(), w, V.data_ptr(), mk, ws.data_ptr(), mk))
        else:
            q.check(lib.qrd_panel_cholqr(st, Pm.data_ptr(), mk, mk, w, tau.data_ptr(), Tm.data_ptr(), w, V.data_ptr(), mk, ws.data_ptr(), mk,
                                         cws.data_ptr(), slabs.data_ptr(), 1 << 22, 0))
        p.sync()
    out.append({"mk": mk, "w": w, "leaf_bytes": 8 * mk * w})
print(json.dumps({"mode": mode, "launches_per_shape": 3, "shapes": out}))
