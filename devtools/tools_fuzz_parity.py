"""Random-shape parity sweep against LAPACK (numpy) on the GPU: sign-normalised R, residual and orthogonality through the device API,
default schedule (CU partition, early look-ahead update, half-size leaf workgroups, first-order second Cholesky factor ...).
python devtools/tools_fuzz_parity.py [seed]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, numpy as np, torch
import cuda_qr_amd as q
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
shapes = [(3000, 2500, 0), (5000, 5000, 256), (10000, 3000, 128), (9000, 4100, 256), (4097, 2049, 64), (6144, 6144, 0), (20000, 2304, 256),
          (12288, 2560, 128), (70000, 320, 0), (33000, 96, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "ragged":        # odd sizes: ragged tiles, leaves narrower than 32, heights not multiples of 4
    shapes = [(1900, 1900, 0), (2500, 2047, 0), (8200, 8190, 0), (12289, 2051, 0), (5000, 1025, 0), (33001, 97, 0), (4099, 4097, 256),
              (6145, 3071, 128), (70001, 321, 0), (20003, 2305, 256)]
if len(sys.argv) > 2 and sys.argv[2] == "edges":         # shapes next to the thresholds of the default rules (block size, look-ahead, partition)
    shapes = [(1536, 1024, 0), (1537, 1024, 0), (1024, 1023, 0), (4243, 4243, 0), (4242, 4242, 0), (32768, 2048, 0), (32767, 2048, 0),
              (10240, 10240, 0), (10239, 10239, 0), (16385, 16384, 0), (8192, 8192, 0), (16384, 1024, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "edges6":        # ... of the round-6 rules (qr_host.c: lookahead_pays, default_blocks, the 32-CU split from 8192^2)
    shapes = [(2828, 2828, 0), (2829, 2829, 0), (3072, 1024, 0), (3073, 1024, 0), (8192, 2048, 0), (8193, 2048, 0), (8193, 4097, 0),
              (20480, 4096, 0), (21000, 4096, 0), (1024, 512, 0), (1024, 511, 0), (1537, 512, 0), (8192, 8191, 0), (10241, 8192, 0),
              (8192, 512, 0), (8193, 512, 0), (4001, 2000, 0), (6143, 2049, 0), (2047, 2047, 0), (777, 555, 0)]
worst = 0.0
for (m, n, nb) in shapes:
    kind = rng.integers(0, 3)
    A = rng.standard_normal((m, n)) if kind == 0 else rng.random((m, n))
    if kind == 2:
        A *= np.logspace(0, -6, n)[None, :]            # graded columns
    p = q.Plan(m, n, nb, 32)
    dA = torch.from_numpy(np.ascontiguousarray(A.T)).cuda(); dtau = torch.zeros(n, dtype=torch.float64, device="cuda")
    dR = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    p.geqrf(dA, m, n, m, dtau); p.extract_r(dA, m, n, m, dR, n, n); p.sync()
    R = np.asfortranarray(dR.cpu().numpy().T)
    Rl = np.linalg.qr(A, mode="r")
    Rn, Rln = O.sign_normalise(R), O.sign_normalise(Rl)
    # column-scaled comparison (graded matrices): relative to each column's norm
    cn = np.linalg.norm(Rln, axis=0)
    err = np.linalg.norm((Rn - Rln) / cn[None, :]) / np.sqrt(n)
    dQ = torch.zeros((n, m), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()          # torch's fill runs on torch's stream, the plan's work on the plan's: order them
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True); p.sync()
    Q = np.asfortranarray(dQ.cpu().numpy().T)
    resid = np.linalg.norm(A - Q @ R) / np.linalg.norm(A)
    orth = np.linalg.norm(Q.T @ Q - np.eye(n))
    print("%6d x %5d nb %3d kind %d  |dR| %.2e  resid %.2e  orth %.2e" % (m, n, nb, kind, err, resid, orth), flush=True)
    worst = max(worst, err)
    assert err < 1e-12 and resid < 1e-13 and orth < 1e-11, (m, n, nb)
    p.close()
print("ok, worst |dR| %.2e" % worst)
