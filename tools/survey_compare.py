"""tools/survey_compare.py a.txt b.txt [c.txt ...] — outputs of tools/patch_survey.py side by side: ms per second of audio per seed and setting,
the ratio of every later setting to the first, medians; as JSON on the last line (profiles/r04_survey.json)."""
import json, re, statistics, sys
runs = []
for path in sys.argv[1:]:
    d = {}
    for ln in open(path):
        m = re.match(r"seed\s+(\d+) B=\s*(\d+) planes=(\d+)\s+([\d.]+) ms/s", ln)
        if m:
            d[int(m.group(1))] = (float(m.group(4)), ln[ln.index("kernel="):].strip() if "kernel=" in ln else "", "regs=" in ln)
    runs.append(d)
seeds = sorted(set.intersection(*[set(r) for r in runs]))
print("seed " + " ".join(f"{p.split('/')[-1][:18]:>18}" for p in sys.argv[1:]) + "   ratios to the first")
ratios = [[] for _ in runs[1:]]
rows = {}
for s in seeds:
    ms = [r[s][0] for r in runs]
    rr = [ms[0] / m for m in ms[1:]]
    for k, x in enumerate(rr):
        ratios[k].append(x)
    rows[s] = ms
    print(f"{s:4d} " + " ".join(f"{m:18.2f}" for m in ms) + "   " + " ".join(f"{x:5.2f}x" for x in rr) + ("  budget" if runs[-1][s][2] else ""))
med = [statistics.median(r[s][0] for s in seeds) for r in runs]
print("median ms/s: " + " ".join(f"{m:.2f}" for m in med) + "; median speed-up per seed: " + " ".join(f"{statistics.median(r):.3f}x" for r in ratios))
print(json.dumps({"settings": sys.argv[1:], "seeds": seeds, "ms_per_s": rows, "median_ms_per_s": med, "median_speedup_vs_first": [statistics.median(r) for r in ratios],
                  "geomean_speedup_vs_first": [statistics.geometric_mean(r) for r in ratios]}))
