"""What runs after the local factorisation has ended in one rank's TSQR step: from a rocprofv3 kernel trace (csv) of
devtools/tools_tsqr_latency.py, the LAST step's kernels from 0.3 ms before the last full-width panel kernel ends to the end of the step.
python devtools/tools_trace_tail.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))) for r in rows), key=lambda x: x[0])
# the last vpass / hr3 / final kernel of the local stream marks the end of the local QR of the last step
last_local = max(i for i, e in enumerate(ev) if "cqr_vpass" in e[2] or "cqr_stream" in e[2])
# walk back to the start of that step's tail: everything within 1.2 ms before the end of the trace
t_end = ev[-1][1]
t0 = t_end - 1_300_000
print("last step: tail of the trace, times in us relative to the end of the last kernel; local-stream reference: last full-width pass kernel ends at %.1f" % ((ev[last_local][1] - t_end) / 1e3))
for s, e, n, q in ev:
    if s >= t0:
        print("%9.1f .. %9.1f  (%6.1f)  q%-4s %s" % ((s - t_end) / 1e3, (e - t_end) / 1e3, (e - s) / 1e3, q, n[:90]))
