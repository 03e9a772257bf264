"""per-step schedule of one look-ahead factorisation from the plan's own HIP-event records:
   python devtools/tools_gantt.py 16384x16384x256"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, torch
import cuda_qr_amd as q
m, n, nb = (int(x) for x in sys.argv[1].split("x"))
p = q.Plan(m, n, nb, 32)
dA = torch.empty((n, m), dtype=torch.float64, device="cuda"); dtau = torch.empty(n, dtype=torch.float64, device="cuda")
for r in range(2):
    p.fill_uniform(dA, m, m, n, seed=12); p.sync()
    p.set_profile(r == 1)
    p.geqrf(dA, m, n, m, dtau); p.sync()
recs = p.get_profile_records()
names = {0: "W.nn", 1: "W.tn", 2: "P", 3: "W.vt", 4: "N", 5: "E"}
# group by step: every class-2 record starts a new panel
step, rows = -1, {}
order = []
for c, a, b in recs:
    if c == 2:
        step += 1
    order.append((step, names[c], a, b))
print("total ms", max(b for _, _, _, b in order))
print("%4s | %-22s | %-22s | %-34s | %-22s" % ("step", "P(s) start..end", "N(s-1) start..end", "W(s-1) vt/tn/nn start..end", "E"))
by = {}
for st, nm, a, b in order:
    by.setdefault(st, {}).setdefault(nm, []).append((a, b))
for st in sorted(by):
    d = by[st]
    f = lambda k: ("%8.2f..%8.2f" % (d[k][0][0], d[k][-1][1])) if k in d else " " * 18
    # the wide update may come in two slices (W1: the next-next panel's columns first, then the rest): first start .. last end
    w0 = min(d[k][0][0] for k in ("W.vt", "W.tn", "W.nn") if k in d) if "W.nn" in d else 0.0
    w1 = d["W.nn"][-1][1] if "W.nn" in d else 0.0
    ws = ("%8.2f..%8.2f (%5.2f)" % (w0, w1, w1 - w0)) if "W.nn" in d else ""
    print("%4d | %s (%5.2f) | %s | %-34s | %s" % (st, f("P"), d["P"][0][1] - d["P"][0][0] if "P" in d else 0, f("N"), ws, f("E")))
p.close()
