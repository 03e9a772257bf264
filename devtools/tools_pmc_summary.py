"""Sum a rocprofv3 --pmc counter per kernel name from the counter_collection CSV."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root: cuda_qr_amd, oracle
import csv, sys, collections
path, counter = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") != counter:
        continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:48]
    tot[name][0] += float(r["Counter_Value"]); tot[name][1] += 1
print(f"{counter}: per-kernel totals (raw counter units) and per-dispatch average")
for k, (v, c) in sorted(tot.items(), key=lambda x: -x[1][0]):
    print(f"{k:50s} dispatches {c:7d} total {v:16.1f} avg {v/c:14.2f}")
