"""The price of a refused full-width panel: geqrf of tall shapes on the plain uniform input and on bench.py --cond's input
(every 128-column panel = its first column + noise / cond), drained per step.  python devtools/tools_cond.py [cond] MxNxNB ..."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time, json
import torch
import cuda_qr_amd as q

def run(m, n, nb, cond, reps=3):
    p = q.Plan(m, n, nb, 32)
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda")
    dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    best = None
    for r in range(reps + 1):
        p.fill_uniform(dA, m, m, n, seed=12)
        p.sync()
        if cond > 0:
            for c in range(0, n, 128):
                dA[c + 1:c + 128] = dA[c:c + 1] + dA[c + 1:c + 128] / cond
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.geqrf(dA, m, n, m, dtau)
        p.sync()
        dt = time.perf_counter() - t0
        if r > 0 and (best is None or dt < best): best = dt
    print(json.dumps({"m": m, "n": n, "nb": nb, "cond": cond, "ms": best * 1e3, "routes": p.route_stats()}), flush=True)
    p.close()

if __name__ == "__main__":
    args = sys.argv[1:]
    cond = float(args.pop(0)) if "x" not in args[0] else 0.0
    for a in args:
        run(*(int(x) for x in a.split("x")), cond)
