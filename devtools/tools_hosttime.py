"""Host issue time vs. device time of one factorisation:  tools_hosttime.py MxNxNB [...]
issue = wall time until qr_geqrf_dev returns (everything queued), total = until the plan's streams have drained."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time, torch
import cuda_qr_amd as q
for arg in sys.argv[1:]:
    m, n, nb = (int(x) for x in arg.split("x"))
    p = q.Plan(m, n, nb, 32)
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    best = None
    for r in range(4):
        p.fill_uniform(dA, m, m, n, seed=12); p.sync()
        t0 = time.perf_counter(); p.geqrf(dA, m, n, m, dtau); t1 = time.perf_counter(); p.sync(); t2 = time.perf_counter()
        if r and (best is None or t2 - t0 < best[1]): best = (t1 - t0, t2 - t0)
    print("%s  issue %.3f ms  total %.3f ms  inct=%s evscope=%s" % (arg, best[0] * 1e3, best[1] * 1e3, os.environ.get("MI355XQR_INCT", "-"),
                                                                  os.environ.get("MI355XQR_EVENT_SCOPE", "-")), flush=True) if (os := _os) else None
    p.close()
