"""Shader clock while the full-width tall panel's passes run (sampler of devtools/probe_clock.hip on its own stream):
python devtools/tools_clock_cqr.py   -- 8 panels of 2097152 x 128 back to back (each pass ~1 ms)"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C, json, time
import numpy as np
import torch
import cuda_qr_amd as q

pc = C.CDLL(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "probe_clock.so"))
pc.clock_probe_launch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int]
pc.clock_probe_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
L = q.lib
q.check(L.qrd_init(), "init")
L.qrd_panel_cqr_ws_doubles.restype = C.c_size_t
L.qrd_panel_cqr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
mk, w = 2097152, 128
ws = torch.zeros(int(L.qrd_panel_cqr_ws_doubles()), dtype=torch.float64, device="cuda")
status = torch.zeros(4, dtype=torch.int32, device="cuda")
A = torch.rand((w, mk), dtype=torch.float64, device="cuda"); V = torch.zeros((w, mk), dtype=torch.float64, device="cuda")
T = torch.zeros((w, w), dtype=torch.float64, device="cuda"); tau = torch.zeros(w, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def work():
    for _ in range(8):
        L.qrd_panel_cqr(None, A.data_ptr(), mk, mk, w, tau.data_ptr(), T.data_ptr(), w, V.data_ptr(), mk, ws.data_ptr(), status.data_ptr())
    L.qrd_device_sync()
work()
st, d = C.c_void_p(), C.c_void_p()
ns = 120
assert pc.clock_probe_launch(C.byref(st), C.byref(d), ns, 300) == 0
t0 = time.perf_counter(); work(); hm = (time.perf_counter() - t0) * 1e3
buf = np.zeros(3 * ns, dtype=np.uint64)
assert pc.clock_probe_collect(st, d, ns, buf.ctypes.data) == 0
a = buf.reshape(ns, 3).astype(np.float64)
ghz = a[:, 0] / a[:, 1] * 0.1
t = (a[:, 2] - a[0, 2]) * 1e-5
busy = t < hm
print(json.dumps({"host_ms": hm, "ghz_mean_while_running": float(ghz[busy].mean()), "ghz_min": float(ghz[busy].min()), "ghz_max": float(ghz[busy].max()),
                  "series_ms_ghz": [[round(float(x), 1), round(float(y), 3)] for x, y in zip(t[busy][::2], ghz[busy][::2])]}))
