"""summarise a rocprofv3 kernel-trace CSV: per-kernel busy time and the idle gaps between consecutive kernels."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = collections.defaultdict(lambda: [0, 0])
gaps = collections.defaultdict(lambda: [0, 0])
prev_end = None; prev_name = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[n][0] += e - s; busy[n][1] += 1
    if prev_end is not None and s > prev_end:
        gaps[n][0] += s - prev_end; gaps[n][1] += 1
    prev_end = max(prev_end or 0, e)
print("span_ms", (t1 - t0) / 1e6, "kernels", len(rows))
for n, (b, c) in sorted(busy.items(), key=lambda x: -x[1][0]):
    g = gaps[n]
    print(f"{n:42s} calls {c:6d} busy {b/1e6:9.3f} ms avg {b/c/1e3:8.2f} us | gap-before total {g[0]/1e6:8.3f} ms avg {g[0]/max(g[1],1)/1e3:6.2f} us")
