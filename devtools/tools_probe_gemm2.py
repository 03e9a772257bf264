"""NN and TN wide-update GEMMs at C3 step shapes on the whole chip (A/B experiments on the kernels)"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import ctypes as C, time, json
import torch
import cuda_qr_amd as q
p = q.Plan(16384, 16384, 256, 32)
m = 16384
A = torch.rand((m, m), dtype=torch.float64, device="cuda"); V = torch.rand((256, m), dtype=torch.float64, device="cuda")
W = torch.empty((m, 256), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for k in (0, 4096, 8192, 12288):
    mk, nt = m - k, m - k - 256
    a2 = A.data_ptr() + 8 * ((k + 256) * m + k)
    res = {}
    for name, fn in (("tn", lambda: p.gemm("T", 256, nt, mk, 1.0, V.data_ptr() + 8 * k, m, a2, m, 0.0, W, 256)),
                     ("nn", lambda: p.gemm("N", mk, nt, 256, -1.0, V.data_ptr() + 8 * k, m, W, 256, 1.0, a2, m))):
        for rep in range(2):
            p.sync(); t0 = time.perf_counter()
            for _ in range(5): fn()
            p.sync(); dt = (time.perf_counter() - t0) / 5
        res[name] = round(2.0 * mk * nt * 256 / dt / 1e12, 2)
    print(json.dumps({"mk": mk, "nt": nt, "tflops": res}), flush=True)
p.close()
