"""MFMA f64 rate against the number of busy CUs (power/DVFS head-room) and GEMM rate against CU-mask width."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import ctypes as C, time, json, sys
import torch
import cuda_qr_amd as q
lib = q.lib
lib.qrd_probe_mfma_f64_point.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 2)()
for blocks in (8, 32, 64, 128, 192, 256, 512, 1024):
    lib.qrd_probe_mfma_f64_point(blocks, 4000, out)
    busy = min(blocks, 256)
    print(json.dumps({"probe": "mfma_f64", "blocks": blocks, "tflops": round(out[0], 2), "per_cu_gflops": round(out[0] * 1e3 / busy, 1),
                      "memtime_ghz": round(out[1], 3), "implied_clock_ghz_at_64cyc": round(out[0] * 1e12 / busy / 4 / 2048 * 64 / 1e9, 3)}), flush=True)
# GEMM NN at the C3 step-0 shape on CU-masked streams
M, N, K = 16128, 15872, 256
A = torch.rand((K, M), dtype=torch.float64, device="cuda")
B = torch.rand((N, K), dtype=torch.float64, device="cuda")
Cm = torch.rand((N, M), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
lib.qrd_stream_create_cumask.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
lib.qrd_stream_sync.argtypes = [C.c_void_p]
for first, count in ((0, 0), (0, 64), (0, 128), (64, 192), (32, 224), (0, 256)):
    st = C.c_void_p()
    q.check(lib.qrd_stream_create_cumask(C.byref(st), first, count))
    for rep in range(2):
        lib.qrd_stream_sync(st)
        t0 = time.perf_counter()
        for _ in range(5):
            q.check(lib.qrd_gemm_nn(st, M, N, K, -1.0, A.data_ptr(), M, B.data_ptr(), K, 1.0, Cm.data_ptr(), M))
        lib.qrd_stream_sync(st)
        dt = (time.perf_counter() - t0) / 5
    print(json.dumps({"probe": "gemm_nn", "cus": count or "unmasked", "first": first, "ms": round(dt * 1e3, 3), "tflops": round(2.0 * M * N * K / dt / 1e12, 2)}), flush=True)
