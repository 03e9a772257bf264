"""ragged tall shapes through qr_geqrf_dev (full-width route for the 128-column panels of >= 196608 rows, leaf chain for the others and
for narrower last panels) against LAPACK: python devtools/tools_cqr_fuzz.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cuda_qr_amd as qr
from gpu_util import dev, host, zeros
worst = 0.0
SHAPES = [(200003, 128, 128, 0), (262147, 384, 128, 0), (300000, 256, 128, 1), (196608, 128, 128, 0), (196607, 128, 128, 0),
          (250001, 130, 128, 0), (230000, 200, 128, 2), (400000, 128, 128, 3)]
if len(_sys.argv) > 1 and _sys.argv[1] == "more":     # the heights the route took over in round 5 (from 8193 rows), ragged widths, every kind
    SHAPES = [(8200, 128, 128, 0), (8321, 256, 128, 3), (9000, 384, 128, 1), (12345, 130, 128, 2), (20000, 512, 128, 0), (33000, 96, 128, 0),
              (65536, 256, 128, 3), (70001, 200, 128, 2), (131072, 256, 128, 1), (16390, 640, 128, 3), (50000, 128, 128, 1), (99999, 257, 128, 0)]
for (m, n, nb, kind) in SHAPES:
    rng = np.random.default_rng(m + n)
    A = rng.random((m, n))
    if kind == 1: A = A * np.logspace(0, -6, n)[None, :]                 # graded columns
    if kind == 2: A[:, 1::2] -= 0.5                                      # mixed signs
    if kind == 3: A = rng.standard_normal((m, n)) @ (np.eye(n) + 0.3 * rng.standard_normal((n, n)))
    p = qr.Plan(m, n, nb, 32)
    dA, dtau, dQ = dev(A), zeros(n, 1), zeros(m, n)
    p.geqrf(dA, m, n, m, dtau); p.sync()
    F = host(dA)
    p.applyq(dA, m, n, m, dtau, dQ, n, m, True); p.sync()
    Q = host(dQ); p.close()
    R = np.triu(F[:n]); Rref = np.linalg.qr(A, mode="r")
    S = np.sign(np.diag(R)) * np.sign(np.diag(Rref))
    cs = np.linalg.norm(Rref, axis=0)
    dR = np.abs((S[:, None] * R - Rref) / cs[None, :]).max()
    resid = np.linalg.norm(A - Q @ R) / np.linalg.norm(A); orth = np.linalg.norm(Q.T @ Q - np.eye(n))
    worst = max(worst, dR)
    print("%8d x %4d nb %3d kind %d  |dR| %.2e  resid %.2e  orth %.2e" % (m, n, nb, kind, dR, resid, orth), flush=True)
    assert dR < 1e-12 and resid < 1e-13 and orth < 1e-11
print("ok, worst |dR| %.2e" % worst)
