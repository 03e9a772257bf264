"""profiles/r02_pmc_traffic.json from the two PMC summaries + the driver's shape list:
python tools_pmc_traffic2.py FETCH_summary.txt WRITE_summary.txt driver_shapes.json <git head>"""
import json, sys, re

def avg(path):
    out = {}
    for l in open(path):
        m = re.match(r"(\S.*?)\s+dispatches\s+(\d+)\s+total\s+([\d.]+)\s+avg\s+([\d.]+)", l)
        if m: out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out

F, Wr, shapes = avg(sys.argv[1]), avg(sys.argv[2]), json.load(open(sys.argv[3]))
steps = shapes["steps"]
nn_alg = sum(s["nn_alg_bytes"] for s in steps) / len(steps)
tn_alg = sum(s["tn_alg_bytes"] for s in steps) / len(steps)
KiB = 1024.0
def find(d, prefix):
    for k in d:
        if k.startswith(prefix): return d[k]
    raise KeyError(prefix)
def entry(prefix, alg):
    f = 2.0 * find(F, prefix)[1] * KiB          # gfx950: FETCH_SIZE counts half of the bytes of coalesced reads
    w = find(Wr, prefix)[1] * KiB
    return {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w,
            "algorithmic_bytes_per_launch": alg, "ratio": (f + w) / alg, "launches": find(F, prefix)[0]}
out = {
 "config": "C3 16384x16384 nb=%d, trailing-update GEMM pair at every 8th outer step (%d launches each), no CU masks" % (shapes["nb"], len(steps)),
 "git_head": sys.argv[4] if len(sys.argv) > 4 else None,
 "nb": shapes["nb"],
 "calibration": {
  "stream_copy_kernel (16 B/lane, 1 GiB read + 1 GiB write per launch)": {
   "FETCH_SIZE_KiB": find(F, "stream_copy_kernel")[1], "true_read_KiB": 1048576, "WRITE_SIZE_KiB": find(Wr, "stream_copy_kernel")[1], "true_write_KiB": 1048576},
  "diff_norm_kernel (8 B/lane, 2 GiB read)": {"FETCH_SIZE_KiB": find(F, "diff_norm_kernel")[1], "true_read_KiB": 2097152},
  "rule": "FETCH_SIZE reports 1/2 of coalesced reads on gfx950 (x2 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE is exact"},
 "gemm_nt_kernel": entry("gemm_nt4_kernel" if any(k.startswith("gemm_nt4_kernel") for k in F) else "gemm_nt_kernel", nn_alg),
 "gemm_nt_kernel_name": next((k for k in F if k.startswith("gemm_nt4_kernel") or k.startswith("gemm_nt_kernel")), None),
 "gemm_tn_kernel<4,4,true,1>": entry("gemm_tn_kernel<4, 4, true, 1>", tn_alg),
}
try:
    out["gemm_tn_wide_kernel<1> (128 x 256 tiles, opt-in)"] = entry("gemm_tn_wide_kernel", tn_alg)
except KeyError:
    pass
json.dump(out, sys.stdout, indent=1)
