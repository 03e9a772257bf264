"""per-queue timeline of one window of a rocprofv3 kernel trace: python tools_trace_timeline.py trace.csv frac window_ms"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]); win = float(sys.argv[3]) * 1e6
rows = [r for r in rows if "fill_uniform" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last geqrf = after the last fill: take everything after the largest idle gap start... simpler: use whole span
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
w0 = t0 + frac * (t1 - t0)
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
sel = [r for r in rows if w0 <= int(r["Start_Timestamp"]) < w0 + win]
queues = collections.Counter(r[qkey] for r in sel)
print("queues in window:", dict(queues))
last_end = {}
for r in sel:
    q = r[qkey]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    g = r.get("Grid_Size", r.get("Grid_Size_X", "?")); wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
    print(f"q{q:>3s} t={((s - w0) / 1e3):9.1f} us dur {((e - s) / 1e3):8.1f} gap {gap:7.1f}  {n}  grid {g}/{wg}")
