"""residual / orthogonality of the guard sweep shapes (tests/test_gpu_multipanel_golden.py::test_cholqr2_guard_threshold_sweep), printed"""
import os as _os, sys as _sys
_ROOT = _os.path.abspath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", ".."))
_sys.path.insert(0, _ROOT)
_sys.path.insert(0, _os.path.join(_ROOT, "tests"))
import numpy as np
import cuda_qr_amd as qr
from gpu_util import dev, host, zeros
shapes = ((2048, 32), (20000, 32), (4096, 128), (16384, 128))
if _os.environ.get("SWEEP_ONLY"): shapes = ((4096, 128), (4096, 64), (8192, 256))
for mk, w in shapes:
    for cond in (1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8):
        rng = np.random.default_rng(int(np.log10(cond)) * 7 + mk)
        U, _ = np.linalg.qr(rng.standard_normal((mk, w)))
        V, _ = np.linalg.qr(rng.standard_normal((w, w)))
        A = (U * np.logspace(0, -np.log10(cond), w)) @ V.T
        p = qr.Plan(mk, w, max(w, 32), 32)
        dA, dtau, dQ, dR = dev(A), zeros(w, 1), zeros(mk, w), zeros(w, w)
        p.geqrf(dA, mk, w, mk, dtau)
        p.extract_r(dA, mk, w, mk, dR, w, w)
        p.applyq(dA, mk, w, mk, dtau, dQ, w, mk, True)
        p.sync()
        R, Q = host(dR), host(dQ)
        st = p.route_stats() if hasattr(p, "route_stats") else {}
        print("%6d x %3d cond %.0e resid %.2e orth %.2e  %s" % (mk, w, cond, np.linalg.norm(A - Q @ R) / np.linalg.norm(A), np.linalg.norm(Q.T @ Q - np.eye(w)), st), flush=True)
        p.close()
