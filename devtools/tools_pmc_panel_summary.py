"""HBM GB/s of the leaf-panel kernels: FETCH_SIZE (x2, gfx950) + WRITE_SIZE per dispatch from two --pmc passes, duration per
dispatch from a --kernel-trace pass of the same driver; dispatches are matched by order within each kernel name."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import csv, sys, collections, json
fetch_csv, write_csv, trace_csv = sys.argv[1:4]
def per_dispatch(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d
F, W = per_dispatch(fetch_csv, "FETCH_SIZE"), per_dispatch(write_csv, "WRITE_SIZE")
T = collections.defaultdict(list)
rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    T[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel                                   dispatch#   read MB  write MB    us     GB/s   (FETCH_SIZE x2 correction applied)")
for k in sorted(F):
    if not any(s in k for s in ("gram32", "cholq", "final2", "final3", "hr2", "hr3", "tsqr_", "hr_top", "slab_reduce")): continue
    n = min(len(F[k]), len(W.get(k, [])), len(T.get(k, [])))
    for i in range(n):
        rd, wr, us = 2 * F[k][i] * 1024 / 1e6, W[k][i] * 1024 / 1e6, T[k][i] / 1e3
        if rd + wr < 1.0: continue
        print(f"{k[:40]:40s} {i:9d} {rd:9.1f} {wr:9.1f} {us:7.1f} {((rd + wr) / 1e3) / (us * 1e-6):8.0f}")
