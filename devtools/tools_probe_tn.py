"""wide TN product (W = (V T)^T A2) alone, whole chip and 192 CUs, with a check against torch"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C, time, json, os
import torch
import cuda_qr_amd as q
lib = q.lib
lib.qrd_gemm_tn_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
lib.qrd_stream_create_cumask.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
lib.qrd_stream_sync.argtypes = [C.c_void_p]
m = 16384
A = torch.rand((m, m), dtype=torch.float64, device="cuda"); V = torch.rand((256, m), dtype=torch.float64, device="cuda")
W = torch.empty((m, 256), dtype=torch.float64, device="cuda"); slabs = torch.empty(16 << 20, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for first, count in ((0, 256), (64, 192)):
    st = C.c_void_p(); q.check(lib.qrd_stream_create_cumask(C.byref(st), first, count))
    for k in (0, 8192, 12288):
        mk, nt = m - k, m - k - 256
        a2 = A.data_ptr() + 8 * ((k + 256) * m + k)
        for rep in range(2):
            lib.qrd_stream_sync(st); t0 = time.perf_counter()
            for _ in range(5):
                q.check(lib.qrd_gemm_tn_update(st, 256, nt, mk, 1.0, V.data_ptr() + 8 * k, m, a2, m, 0.0, W.data_ptr(), 256, slabs.data_ptr(), slabs.numel()))
            lib.qrd_stream_sync(st); dt = (time.perf_counter() - t0) / 5
        err = None
        if mk <= 4096:
            want = A[k + 256:, k:].double() @ V[:, k:].T            # (nt x mk) @ (mk x 256): W^T stored as (nt, 256)
            err = float((W[:nt] - want).abs().max())
        print(json.dumps({"bk32": os.environ.get("MI355XQR_TN_BK32", "0"), "cus": count, "mk": mk, "nt": nt, "ms": round(dt * 1e3, 3),
                          "tflops": round(2.0 * mk * nt * 256 / dt / 1e12, 2), "maxerr": err}), flush=True)
