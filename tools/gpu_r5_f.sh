#!/bin/bash
# Round 5, pass F: is the soaks' transient the copy?  (tools/copy_probe.py: 16 processes, the same device bytes copied back again and again);
# the suite; P3 with and without the scalar track-change gating (one box, alternating); the flagged seed 66697 under the new motion rule; bench
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( timeout 900 python tools/copy_probe.py 1500 16 ) > $OUT/f_copy_probe.log 2>&1; echo "== copy probe rc=$?"; grep -c differ $OUT/f_copy_probe.log; grep "copy " $OUT/f_copy_probe.log | head -12 | cut -c1-250; grep "^worker.*copies of" $OUT/f_copy_probe.log | head -16
( timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 1200 ) > $OUT/f_tests.log 2>&1; echo "== tests rc=$?"; tail -6 $OUT/f_tests.log | cut -c1-300
( SOAK_VT=200,6000 DBG_FLAGS=0,2,4 timeout 300 python tools/dbg_default.py 66697 ) > $OUT/f_66697.log 2>&1; grep "^seed" $OUT/f_66697.log | cut -c1-200
cp s-rack_amd/libsrack_hip.so /tmp/_new.so
( bash tools/ab.sh "--no-side-configs --workload p3" s-rack_amd/libsrack_hip_prev.so /tmp/_new.so ) > $OUT/f_ab_p3.log 2>&1; echo "== A/B P3 (prev, new)"; cat $OUT/f_ab_p3.log | tail -7
cp /tmp/_new.so s-rack_amd/libsrack_hip.so
( timeout 900 python bench.py ) > $OUT/f_bench.json 2> $OUT/f_bench.err; echo "== bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/f_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("cfg3 %.3f ms frac %.4f kernel %.4f" % (d["ms_per_step"], r["frac"], r["frac_kernel"]))
for k in sorted(r):
    if k.endswith("_ms_per_step"): print("  ", k, round(r[k],3), "frac_hbm", round(r.get(k.replace("_ms_per_step","_frac_hbm"),0),4))
PY
