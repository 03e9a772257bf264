"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): calibration kernels with known
byte counts in the access widths the GEMMs use, then one C3 factorisation (profiles/README.md)."""
import sys
import torch
import cuda_qr_amd as q

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
p = q.Plan(m, n, nb, 32)
A = torch.empty((n, m), dtype=torch.float64, device="cuda")
tau = torch.empty(n, dtype=torch.float64, device="cuda")
q.probe_copy_gbps()                                   # stream_copy_kernel: 6 launches x (1 GiB read + 1 GiB write), 16 B/lane
p.fill_uniform(A, m, m, n, seed=12)                   # fill_uniform_kernel: m*n*8 bytes written, 8 B/lane
p.sync()
p.diffnorm(A, m, m, n, seed=12)                       # diff_norm_kernel: m*n*8 bytes read, 8 B/lane
p.geqrf(A, m, n, m, tau)
p.sync()
print("done", m, n, nb)
