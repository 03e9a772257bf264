"""raw HIP-event records of one look-ahead factorisation: python devtools/tools_records.py 16384x16384x256 t0_ms t1_ms"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, torch
import cuda_qr_amd as q
m, n, nb = (int(x) for x in sys.argv[1].split("x"))
lo, hi = float(sys.argv[2]), float(sys.argv[3])
p = q.Plan(m, n, nb, 32)
dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
for r in range(2):
    p.fill_uniform(dA, m, m, n, seed=12); p.sync()
    p.set_profile(r == 1)
    p.geqrf(dA, m, n, m, dtau); p.sync()
recs = p.get_profile_records()
names = {0: "W.nn", 1: "W.tn", 2: "P", 3: "W.vt", 4: "N", 5: "E"}
print("total ms", max(b for _, _, b in recs))
for c, a, b in sorted(recs, key=lambda r: r[1]):
    if lo <= a <= hi:
        print("%-5s %9.3f .. %9.3f  (%6.3f)" % (names[c], a, b, b - a))
p.close()
