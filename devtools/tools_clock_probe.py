"""sample the SMI clocks while a long GEMM / MFMA loop is running (is fp64 MFMA clock-throttled?)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import subprocess, threading, time, json, ctypes as C
import torch
import cuda_qr_amd as q
lib = q.lib
M, N, K = 16128, 15872, 256
A = torch.rand((K, M), dtype=torch.float64, device="cuda"); B = torch.rand((N, K), dtype=torch.float64, device="cuda")
Cm = torch.rand((N, M), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def smi(tag):
    for cmd in (["rocm-smi", "--showclocks", "--showpower"], ):
        try:
            o = subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout
            keep = [l.strip() for l in o.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l or "fclk" in l)]
            print(tag, " | ".join(keep)[:600], flush=True)
        except Exception as e:
            print(tag, "smi failed", e)
smi("idle")
p = q.Plan(1024, 1024)
st = p.stream
def load(kind, secs):
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        if kind == "gemm":
            for _ in range(50):
                lib.qrd_gemm_nn(st, M, N, K, -1.0, A.data_ptr(), M, B.data_ptr(), K, 1.0, Cm.data_ptr(), M)
            p.sync(); n += 50
        else:
            out = (C.c_double * 2)()
            lib.qrd_probe_mfma_f64_point(1024, 16000, out); n += 1
    print(kind, "launches", n, flush=True)
lib.qrd_probe_mfma_f64_point.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
for kind in ("gemm", "mfma"):
    th = threading.Thread(target=load, args=(kind, 8.0)); th.start()
    time.sleep(2.0); smi(kind + "@2s"); time.sleep(1.0); smi(kind + "@4s+")
    th.join()
