"""Shader clock while the library's kernels run: a one-wave sampler on its own stream (devtools/probe_clock.hip: s_memtime ticks per
s_memrealtime tick over ~1 ms sleeps) next to (a) nothing, (b) two 16384^2 factorisations (look-ahead schedule: update GEMMs on 224 CUs
+ panel chain), (c) the pure-MFMA probe.   python devtools/tools_clock_under_load.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C, json, time
import numpy as np
import torch
import cuda_qr_amd as q

pc = C.CDLL(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "probe_clock.so"))
pc.clock_probe_launch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int]
pc.clock_probe_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]


def series(fn, nsamples, sleeps=300):
    st, d = C.c_void_p(), C.c_void_p()
    assert pc.clock_probe_launch(C.byref(st), C.byref(d), nsamples, sleeps) == 0
    t0 = time.perf_counter()
    fn()
    host_ms = (time.perf_counter() - t0) * 1e3
    buf = np.zeros(3 * nsamples, dtype=np.uint64)
    assert pc.clock_probe_collect(st, d, nsamples, buf.ctypes.data) == 0
    a = buf.reshape(nsamples, 3).astype(np.float64)
    ghz = a[:, 0] / a[:, 1] * 0.1
    t = (a[:, 2] - a[0, 2]) * 1e-5          # ms
    return t, ghz, host_ms


m = n = 16384
p = q.Plan(m, n, 256, 32)
dA = torch.empty((n, m), dtype=torch.float64, device="cuda")
dtau = torch.empty(n, dtype=torch.float64, device="cuda")
p.fill_uniform(dA, m, m, n, seed=12); p.sync()
p.geqrf(dA, m, n, m, dtau); p.sync()         # warm-up
p.fill_uniform(dA, m, m, n, seed=12); p.sync()


def work_qr():
    p.geqrf(dA, m, n, m, dtau)
    p.sync()


out = {}
t, g, _ = series(lambda: time.sleep(0.05), 40)
out["idle"] = {"ghz_mean": float(g.mean()), "ghz_min": float(g.min()), "ghz_max": float(g.max())}
t, g, hm = series(work_qr, 160)
busy = t < hm                                 # samples that started while the factorisation was running (host clock: issue + sync)
out["qr_16384"] = {"host_ms": hm, "ghz_mean_while_running": float(g[busy].mean()), "ghz_min": float(g[busy].min()),
                   "ghz_first_100ms_mean": float(g[(t < min(hm, 100.0))].mean()),
                   "series_ms_ghz": [[round(float(a), 1), round(float(b), 3)] for a, b in zip(t[::4], g[::4])]}
t, g, hm = series(lambda: q.probe_mfma_f64_tflops(), 60)
busy = t < hm
out["mfma_probe"] = {"host_ms": hm, "ghz_mean_while_running": float(g[busy].mean()), "ghz_min": float(g[busy].min())}
print(json.dumps(out))
p.close()
