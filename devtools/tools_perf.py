"""ad-hoc perf exploration (not the bench contract): time geqrf with per-class event profile."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import sys, time, json, os
import torch
import cuda_qr_amd as q

def run(m, n, nb, ib=32, reps=2):
    p = q.Plan(m, n, nb, ib)
    if nb == 0: nb = q.default_block_size(m, n)[0]      # MxNx0: the block size (and schedule) the library picks for the shape
    dA = torch.empty((n, m), dtype=torch.float64, device="cuda")
    dtau = torch.empty(n, dtype=torch.float64, device="cuda")
    best = None
    for r in range(reps + 1):
        p.fill_uniform(dA, m, m, n, seed=12)
        p.sync()
        p.set_profile(r == reps)
        t0 = time.perf_counter()
        p.geqrf(dA, m, n, m, dtau)
        p.sync()
        dt = time.perf_counter() - t0
        if r > 0 and (best is None or dt < best) and r < reps: best = dt
        if r == reps:
            prof = p.get_profile(); tprof = dt
    fl = q.flops(m, n)
    line = {"m": m, "n": n, "nb": nb, "ib": ib, "ms": best * 1e3, "tflops": fl / best / 1e12, "ms_profiled": tprof*1e3}
    for k, v in prof.items():
        if v["launches"]:
            line[k] = {"ms": round(v["ms"], 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "n": v["launches"]}
    if os.environ.get("CHECK"):
        z = lambda r, c: torch.zeros((c, r), dtype=torch.float64, device="cuda")
        for rep in range(int(os.environ["CHECK"])):
            p.fill_uniform(dA, m, m, n, seed=12)
            dQ, dR, dQR = z(m, n), z(n, n), z(m, n)
            torch.cuda.synchronize()
            p.geqrf(dA, m, n, m, dtau)
            p.extract_r(dA, m, n, m, dR, n, n)
            p.applyq(dA, m, n, m, dtau, dQ, n, m, True)
            p.gemm("N", m, n, n, 1.0, dQ, m, dR, n, 0.0, dQR, m)
            p.sync()
            d, a = p.diffnorm(dQR, m, m, n, seed=12)
            line.setdefault("resid", []).append(float((d / a) ** 0.5))
    print(json.dumps(line), flush=True)
    p.close()

if __name__ == "__main__":
    cfgs = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    for c in cfgs:
        run(*c)
