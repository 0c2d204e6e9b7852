"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 output) into small files under profiles/ that can be committed."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
out = {"tag": tag, "kernels": {}, "counters": {}}


def find(sub, pattern):
    hits = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(stats).read())
    for r in rows:
        out["kernels"][r["Name"][:120]] = {k: r[k] for k in r if k != "Name"}

for sub in ("pmc_sq", "pmc_wr", "pmc_rd", "pmc_mem"):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    acc = {}
    for r in csv.DictReader(open(cc)):
        name = r.get("Kernel_Name", "")[:60]
        key = (name, r.get("Counter_Name"))
        a = acc.setdefault(key, [0.0, 0])
        a[0] += float(r.get("Counter_Value", 0))
        a[1] += 1
    for (name, counter), (total, n) in sorted(acc.items()):
        out["counters"].setdefault(name, {})[counter] = {"sum_over_dispatches": total, "dispatches": n, "per_dispatch": total / max(n, 1)}

json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
