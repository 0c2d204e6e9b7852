#!/bin/bash
# Round 5, pass D: the suite and the bench line after a change to the kernels
set -u
OUT=gpurun_out/r5; mkdir -p $OUT; TAG=${1:-d}
( timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 1200 ) > $OUT/${TAG}_tests.log 2>&1; echo "== tests rc=$?"; tail -12 $OUT/${TAG}_tests.log | cut -c1-300
( timeout 900 python bench.py ) > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "== bench rc=$?"; TAG=$TAG python - <<'PY'
import json, os
d=json.loads(open("gpurun_out/r5/%s_bench.json" % os.environ["TAG"]).read().strip().splitlines()[-1])
r=d["roofline"]
print("cfg3 %.3f ms frac %.4f kernel %.4f" % (d["ms_per_step"], r["frac"], r["frac_kernel"]))
for k in sorted(r):
    if k.endswith("_ms_per_step"): print("  ", k, round(r[k],3), "frac_hbm", round(r.get(k.replace("_ms_per_step","_frac_hbm"),0),4))
PY
