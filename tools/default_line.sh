#!/bin/bash
# the default bench line exactly as the driver runs it (python bench.py), with its wall time and a digest
mkdir -p gpurun_out/r4
( time python bench.py ) > gpurun_out/r4/default_line.json 2> gpurun_out/r4/default_line.err
grep -E "^real" gpurun_out/r4/default_line.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/default_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step", round(d["ms_per_step"], 3), "frac", round(r["frac"], 4), "frac_kernel", round(r["frac_kernel"], 4))
for k in sorted(r):
    if k.startswith("cfg"):
        print("  ", k, r[k] if not isinstance(r[k], float) else round(r[k], 4))
c = d["cpu_baseline"]
print("cpu_baseline", c["value"], c["cores"], c["single_thread_value"], c["single_thread_sample"])
for k, v in d["configs"].items():
    print("  ", k, v["program"][-80:])
PY
