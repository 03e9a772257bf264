"""HBM bytes of a WHOLE factorisation from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over devtools/tools_one.py:
python tools_pmc_total.py FETCH_counter_collection.csv WRITE_counter_collection.csv m n [factorisations in the run]
Sums every dispatch except the input generator (fill_uniform_kernel) and divides by the number of factorisations.
gfx950: FETCH_SIZE counts half of the bytes of coalesced reads (x2, calibrated in profiles/r03_pmc_traffic.json); both in KiB."""
import csv, sys, json, collections
f, w, m, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 1
def per_kernel(path, counter):
    t = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter: continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:44]
        t[name][0] += float(r["Counter_Value"]) * 1024.0; t[name][1] += 1
    return t
F, W = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
rows = []
for k in sorted(set(F) | set(W)):
    if k.startswith("fill_uniform") or k.startswith("__amd_rocclr"): continue
    rd, wr = 2.0 * F.get(k, [0, 0])[0] / runs, W.get(k, [0, 0])[0] / runs
    rows.append((rd + wr, k, rd, wr, max(F.get(k, [0, 0])[1], W.get(k, [0, 0])[1]) // runs))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
alg = 16.0 * m * n
print(json.dumps({"m": m, "n": n, "factorisations_in_run": runs, "hbm_bytes_per_factorisation": tot, "algorithmic_bytes_16mn": alg,
                  "ratio": tot / alg, "method": "2*FETCH_SIZE + WRITE_SIZE (KiB) summed over every dispatch of the factorisation, separate --pmc passes",
                  "per_kernel": [{"kernel": k, "read_bytes": rd, "written_bytes": wr, "launches": c} for _, k, rd, wr, c in rows[:14]]}, indent=1))
