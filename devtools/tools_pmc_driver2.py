"""Workload for the round-2 PMC passes (rocprofv3 --pmc ...), see profiles/README.md.

rocprofv3 counter collection crashes on this pool beyond ~16k dispatches per process and on CU-masked streams, so the counters
are taken (MI355XQR_SPLIT=0: no CU masks) on (a) calibration kernels with known byte counts and (b) the wide trailing-update
GEMM pair at the exact C3 shapes of every 8th outer step (mk = nt + nb = 16384 - k, K = nb), launched through the update's own
launch helpers: gemm_tn_kernel<4,4,true,1> (+ slab reduce) producing Wt = A2^T (V T) and gemm_nt_kernel<true,0> doing
A2 -= V Wt^T -- the same kernels, tiles and split-K the factorisation uses."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C
import json
import sys

import torch

import cuda_qr_amd as q

m = n = 16384
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = q.Plan(m, n, nb, 32)
A = torch.empty((n, m), dtype=torch.float64, device="cuda")
V = torch.empty((nb, m), dtype=torch.float64, device="cuda")
Wt = torch.empty((nb, n), dtype=torch.float64, device="cuda")
q.probe_copy_gbps()                                   # stream_copy_kernel: 6 x (1 GiB read + 1 GiB write), 16 B/lane
p.fill_uniform(A, m, m, n, seed=12)                   # fill_uniform_kernel: m*n*8 B written, 8 B/lane
p.fill_uniform(V, m, m, nb, seed=13)
p.sync()
p.diffnorm(A, m, m, n, seed=12)                       # diff_norm_kernel: m*n*8 B read, 8 B/lane
lib = q.lib
lib.qrd_gemm_tn_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
st = p.stream
slabs = torch.empty(16 << 20, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
shapes = []
for k in range(0, n - nb, 8 * nb):
    mk, nt = m - k, n - k - nb
    a2 = A.data_ptr() + 8 * ((k + nb) * m + k)
    v = V.data_ptr() + 8 * k
    q.check(lib.qrd_gemm_tn_update(st, nt, nb, mk, 1.0, a2, m, v, m, 0.0, Wt.data_ptr(), nt, slabs.data_ptr(), slabs.numel()))   # Wt = A2^T (V T)
    if hasattr(lib, "qrd_gemm_tn_update_wide"):      # round 3: the same product on 128 x 256 workgroup tiles (A2 read half as often)
        q.check(lib.qrd_gemm_tn_update_wide(st, nt, nb, mk, 1.0, a2, m, v, m, 0.0, Wt.data_ptr(), nt, slabs.data_ptr(), slabs.numel()))
    q.check(lib.qrd_gemm_nt(st, mk, nt, nb, -1, v, m, Wt.data_ptr(), nt, a2, m, -1, None))                                        # A2 -= V Wt^T
    shapes.append({"k": k, "mk": mk, "nt": nt, "nb": nb, "nn_alg_bytes": 16 * mk * nt + 8 * mk * nb + 8 * nb * nt,
                   "tn_alg_bytes": 8 * mk * (nt + nb), "flops_each": 2 * mk * nt * nb})
p.sync()
print(json.dumps({"nb": nb, "steps": shapes}))
