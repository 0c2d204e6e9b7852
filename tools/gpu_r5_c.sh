#!/bin/bash
# Round 5, pass C: the suite with single oscillators exact (config 4's modulator), the bench line, the horizon curves; then the one-second soak again,
# 16 processes on the device, with an immediate retry of whatever differs (is the transient seen in passes A and B2 the device's or the oracle's?)
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 1200 ) > $OUT/c_tests.log 2>&1; echo "== tests rc=$?"; tail -12 $OUT/c_tests.log | cut -c1-300
( timeout 900 python bench.py ) > $OUT/c_bench.json 2> $OUT/c_bench.err; echo "== bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/c_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("cfg3 %.3f ms frac %.4f kernel %.4f" % (d["ms_per_step"], r["frac"], r["frac_kernel"]))
for k in sorted(r):
    if k.endswith("_ms_per_step"): print("  ", k, round(r[k],3), "frac_hbm", round(r.get(k.replace("_ms_per_step","_frac_hbm"),0),4))
PY
( HORIZON_WORKLOADS=cfg4,cfg4_b1024 HORIZON_FLAGS=0,32,34,1,64 timeout 900 python tools/horizon.py $OUT/horizon_cfg4.json ) > $OUT/c_horizon.log 2>&1; echo "== horizon rc=$?"; cat $OUT/c_horizon.log | cut -c1-230
( SOAK_RETRY=1 SOAK_VT=16,48000 SOAK_TIMEOUT=1100 timeout 1200 python tools/soak_par.py retry_1s 56000 60000 16 ) > $OUT/c_soak_retry.log 2>&1; echo "== soak retry rc=$?"; tail -8 $OUT/c_soak_retry.log | cut -c1-230
grep -h "^RETRY" $OUT/soak_retry_1s_w*.log | cut -c1-420 | head -40
