"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 rocpd SQLite output) into small committed files under profiles/.

    python profiles/summarize.py r01

Writes profiles/<tag>_kernel_stats.csv (the --stats view: per-kernel calls / total / average duration)
and profiles/<tag>_summary.json (stats + PMC counters per kernel, per dispatch, with the derived HBM bytes).
WRITE_SIZE / FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM) — the raw values are kept and the corrected figure is given separately.
"""
import csv
import glob
import json
import os
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
out = {"tag": tag, "kernel_stats_us": {}, "counters_per_dispatch": {}, "derived": {}}


def db(sub):
    hits = glob.glob(os.path.join(src, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(hits[0]) if hits else None


con = db("trace")
if con:
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow(r)
            out["kernel_stats_us"][r[0]] = {"calls": r[1], "total": r[2], "average": r[3], "percent": r[4]}

for sub in ("pmc_sq", "pmc_wr", "pmc_rd", "pmc_mem"):
    con = db(sub)
    if not con:
        continue
    q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
    for name, counter, n, avg in con.execute(q):
        out["counters_per_dispatch"].setdefault(name, {})[counter] = {"dispatches": n, "avg": avg}

for name, c in out["counters_per_dispatch"].items():
    d = {}
    if "WRITE_SIZE" in c:
        d["hbm_write_bytes"] = c["WRITE_SIZE"]["avg"] * 1024
    if "FETCH_SIZE" in c:
        d["hbm_read_bytes_raw"] = c["FETCH_SIZE"]["avg"] * 1024
        d["hbm_read_bytes_gfx950_corrected"] = c["FETCH_SIZE"]["avg"] * 1024 * 2
    if "hbm_write_bytes" in d and "hbm_read_bytes_gfx950_corrected" in d:
        d["hbm_traffic_bytes"] = d["hbm_write_bytes"] + d["hbm_read_bytes_gfx950_corrected"]
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"]["avg"] / c["SQ_WAVES"]["avg"]
        d["salu_insts_per_wave"] = c["SQ_INSTS_SALU"]["avg"] / c["SQ_WAVES"]["avg"]
    if d:
        out["derived"][name] = d

json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out["kernel_stats_us"], indent=1))
print(json.dumps(out["derived"], indent=1))
