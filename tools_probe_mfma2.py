import ctypes as C, json
import torch
import cuda_qr_amd as q
lib = q.lib
lib.qrd_probe_mfma_f64_grid.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
t = C.c_double()
for blocks in (256, 512, 1024):
    q.check(lib.qrd_probe_mfma_f64_grid(blocks, 4000, C.byref(t)))
    print(json.dumps({"probe": "mfma_f64_grid4x4", "blocks": blocks, "tflops": round(t.value, 2),
                      "cycles_per_mfma_at_2.39GHz": round(1024 * min(blocks,1024)/1024 * 2048 * 2.39e9 / (t.value * 1e12) * (1024/min(blocks*4,1024) if blocks<256 else 1), 1)}), flush=True)
