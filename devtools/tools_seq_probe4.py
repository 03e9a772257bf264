"""how many extra live (idle) streams in the process does it take to slow a newly created two-stream plan?"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import time, ctypes as C, torch
import cuda_qr_amd as q
lib = q.lib
lib.qrd_stream_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]; lib.qrd_stream_create.restype = C.c_int
lib.qrd_stream_create_cumask.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]; lib.qrd_stream_create_cumask.restype = C.c_int
def t(m, n, nb, reps=3):
    p = q.Plan(m, n, nb, 32)
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    best = 1e30
    for r in range(reps + 1):
        p.fill_uniform(dA, m, m, n, seed=12); p.sync()
        t0 = time.perf_counter(); p.geqrf(dA, m, n, m, dtau); p.sync(); dt = time.perf_counter() - t0
        if r: best = min(best, dt)
    p.close(); del dA, dtau
    return best * 1e3
print("no extra streams             8192^2 %.2f ms" % t(8192, 8192, 256))
extra = []
for k in (1, 2, 3, 4, 6, 8, 12):
    while len(extra) < k:
        h = C.c_void_p(); assert lib.qrd_stream_create(C.byref(h), 0) == 0; extra.append(h)
    print("%2d extra plain idle streams  8192^2 %.2f ms" % (k, t(8192, 8192, 256)))
for j in range(2):
    h = C.c_void_p(); assert lib.qrd_stream_create_cumask(C.byref(h), 0, 32) == 0; extra.append(h)
print("+ 2 extra masked idle streams 8192^2 %.2f ms   16384^2 %.2f ms" % (t(8192, 8192, 256), t(16384, 16384, 256)))
