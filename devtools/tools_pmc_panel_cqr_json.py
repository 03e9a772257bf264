"""profiles/r05_pmc_panel_hbm.json from the two PMC summaries of devtools/rounds/r5/scripts_r5_pmc.sh over tools_cqr_perf.py 262144 128 0:
python devtools/tools_pmc_panel_cqr_json.py cqr_FETCH_SIZE_summary.txt cqr_WRITE_SIZE_summary.txt <git head>"""
import json, re, sys

def avg(path):
    out = {}
    for l in open(path):
        m = re.match(r"(\S.*?)\s+dispatches\s+(\d+)\s+total\s+([\d.]+)\s+avg\s+([\d.]+)", l)
        if m: out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out

F, W = avg(sys.argv[1]), avg(sys.argv[2])
mk, w = 262144, 128
kern, tot = {}, 0.0
npanel = max(float(n) for name, (n, f) in F.items() if "cqr_chol_kernel" in name)     # one Cholesky launch per panel
for name in F:
    if "cqr_" not in name: continue
    short = name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").strip()
    n, f = F[name]; wn, wr = W.get(name, (n, 0.0))
    per_panel = n / npanel                       # (the reduce kernel runs twice per panel)
    rd, wt = 2.0 * f * 1024 * per_panel, wr * 1024 * per_panel
    kern[short] = {"read_MB": round(rd / 1e6, 1), "write_MB": round(wt / 1e6, 1), "launches_per_panel": per_panel}
    tot += rd + wt
alg = 16.0 * mk * w
print(json.dumps({"mk": mk, "w": w, "hbm_bytes_per_panel": tot, "algorithmic_bytes_16_mk_w": alg, "ratio": tot / alg, "kernels": kern,
                  "covers": "every kernel of one full-width panel (qrd_panel_cqr): Gram pass + reduce, Cholesky, Q + G2 pass + reduce, reconstruction (LU, post), V pass, top block",
                  "method": "2*FETCH_SIZE + WRITE_SIZE per dispatch (KiB, gfx950 correction), separate rocprofv3 --pmc passes over devtools/tools_cqr_perf.py 262144 128 0; devtools/rounds/r5/scripts_r5_pmc.sh",
                  "git_head": sys.argv[3] if len(sys.argv) > 3 else None}, indent=1))
