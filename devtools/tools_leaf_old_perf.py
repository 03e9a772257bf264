"""per-leaf time of the per-leaf launch sequence (old path) inside whole single-stream factorisations of m x n, nb: panel ms / leaves"""
import os, sys, json, subprocess
shapes = [(512, 512, 64), (1024, 1024, 64), (2048, 2048, 64), (3072, 3072, 64), (4096, 4096, 64), (2048, 2048, 128), (4096, 4096, 128)]
for fused in ("0", "1"):
    env = dict(os.environ, MI355XQR_FUSED_PANEL=fused, MI355XQR_FUSED_MIN_ROWS="0", MI355XQR_LOOKAHEAD="0")
    out = subprocess.run([sys.executable, "devtools/tools_perf.py"] + [f"{m}x{n}x{nb}" for m, n, nb in shapes], env=env, capture_output=True, text=True).stdout
    for l in out.splitlines():
        try:
            d = json.loads(l)
        except Exception:
            continue
        leaves = d["n"] // 32
        print(f"fused={fused} {d['m']}x{d['n']} nb {d['nb']}: {d['ms']:.2f} ms, panel {d['panel']['ms']:.2f} ms = {d['panel']['ms'] * 1e3 / leaves:.1f} us per leaf (avg over all heights)")
