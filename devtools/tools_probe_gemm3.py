"""wide-update NN kernel: 4-wave vs 8-wave form (MI355XQR_NN_WAVES) on C3 step shapes, whole chip and 192 CUs, with a check"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import ctypes as C, time, json, os
import torch
import cuda_qr_amd as q
lib = q.lib
lib.qrd_gemm_nn_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_double, C.c_void_p, C.c_int]
lib.qrd_stream_create_cumask.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
lib.qrd_stream_sync.argtypes = [C.c_void_p]
K = 256
for (M, N) in ((16128, 15872), (8192, 7936), (4096, 3840), (4000, 3000)):
    A = torch.rand((K, M), dtype=torch.float64, device="cuda"); B = torch.rand((N, K), dtype=torch.float64, device="cuda")
    C0 = torch.rand((N, M), dtype=torch.float64, device="cuda")
    for first, count in ((0, 256), (64, 192)):
        st = C.c_void_p(); q.check(lib.qrd_stream_create_cumask(C.byref(st), first, count))
        Cm = C0.clone(); torch.cuda.synchronize()
        q.check(lib.qrd_gemm_nn_update(st, M, N, K, -1.0, A.data_ptr(), M, B.data_ptr(), K, 1.0, Cm.data_ptr(), M))
        lib.qrd_stream_sync(st)
        err = float((Cm - (C0 - B @ A)).abs().max()) if M <= 4096 else None
        for rep in range(2):
            lib.qrd_stream_sync(st); t0 = time.perf_counter()
            for _ in range(5):
                q.check(lib.qrd_gemm_nn_update(st, M, N, K, -1.0, A.data_ptr(), M, B.data_ptr(), K, 1.0, Cm.data_ptr(), M))
            lib.qrd_stream_sync(st); dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"waves": os.environ.get("MI355XQR_NN_WAVES", "8"), "shape": [M, N, K], "cus": count, "ms": round(dt * 1e3, 3),
                          "tflops": round(2.0 * M * N * K / dt / 1e12, 2), "maxerr": err}), flush=True)
    del A, B, C0, Cm
