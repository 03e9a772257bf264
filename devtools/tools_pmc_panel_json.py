"""profiles/r02_pmc_panel_hbm.json from the per-dispatch table of devtools/rounds/r2/scripts_r2_pmc_panel.sh (cholqr_hbm_summary.txt):
the three streaming kernels of a 262144 x 32 CholeskyQR2 leaf, averaged over the dispatches that read >= 60 MB.
python devtools/tools_pmc_panel_json.py gpurun_out/pmc_panel_r02/cholqr_hbm_summary.txt <git head>"""
import sys, json, collections
rows = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) < 6 or not p[-1].replace(".", "").isdigit():
        continue
    try:
        gbps, us, wr, rd = float(p[-1]), float(p[-2]), float(p[-3]), float(p[-4])
    except ValueError:
        continue
    name = " ".join(p[:-5])
    if rd >= 60.0:
        rows[name].append((rd, wr, us))
out = {"leaf": "262144 x 32 (67.1 MB), CholeskyQR2 leaf as the tall-skinny plans run it (gram32 / [chol1] cholq4_tall / final3), dispatches averaged",
       "git_head": sys.argv[2] if len(sys.argv) > 2 else None, "kernels": {}}
tot_b = tot_us = 0.0
for name, v in rows.items():
    if not any(k in name for k in ("gram32", "cholq2", "cholq4", "final3")):
        continue
    rd = sum(x[0] for x in v) / len(v); wr = sum(x[1] for x in v) / len(v); us = sum(x[2] for x in v) / len(v)
    out["kernels"][name] = {"read_MB": round(rd, 1), "write_MB": round(wr, 1), "us": round(us, 1), "GBps": round((rd + wr) / us * 1e3)}
    tot_b += (rd + wr) * 1e6; tot_us += us
out["hbm_bytes_per_leaf_streaming_kernels"] = tot_b
out["us_per_leaf_streaming_kernels"] = tot_us
out["algorithmic_bytes_per_leaf"] = 2 * 262144 * 32 * 8
out["note"] = ("inside a factorisation only the first leaf of an outer panel runs gram32_kernel: the others get their Gram matrix from the "
               "previous leaf's in-panel update (leaf_update_gram_kernel), whose bytes are not in this table")
out["method"] = ("2*FETCH_SIZE + WRITE_SIZE per dispatch (gfx950 correction), duration from a --kernel-trace pass of the same driver; "
                 "devtools/rounds/r2/scripts_r2_pmc_panel.sh")
print(json.dumps(out, indent=1))
