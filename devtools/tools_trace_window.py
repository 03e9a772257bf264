"""Kernel-by-kernel view of a few steps of a factorisation from a rocprofv3 kernel trace (csv): everything between the starts of the
i-th and j-th LAST launches of a marker kernel (default panel_fused_kernel), times in us from the window's start, one line per kernel
with its queue.  python devtools/tools_trace_window.py <kernel_trace.csv> [i j [marker]]   (i > j: "12 10" = two steps, eleven from the end)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
i, j = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 10)
marker = sys.argv[4] if len(sys.argv) > 4 else "panel_fused_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", ""))) for r in rows), key=lambda x: x[0])
marks = [k for k, e in enumerate(ev) if marker in e[2]]
a, b = ev[marks[-i]][0], ev[marks[-j]][0]
print("window: %s launch -%d .. -%d, %.1f us" % (marker, i, j, (b - a) / 1e3))
for s, e, n, q in ev:
    if a <= s < b:
        print("%9.1f .. %9.1f  (%6.1f)  q%-3s %s" % ((s - a) / 1e3, (e - a) / 1e3, (e - s) / 1e3, q, n.replace("(anonymous namespace)::", "")[:100]))
