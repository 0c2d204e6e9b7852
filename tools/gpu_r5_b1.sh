#!/bin/bash
# Round 5, pass B1: the -m gpu suite under the derived error bound; a soak with every scratch buffer poisoned (does anything read a word
# nobody wrote?); the error-over-time curves (config 4's fast kernels beside its new default); the flagship's speculative tile A/B; the bench line
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 1200 ) > $OUT/b1_tests.log 2>&1; echo "== tests rc=$?"; tail -15 $OUT/b1_tests.log | cut -c1-300
cp s-rack_amd/libsrack_hip.so /tmp/_new.so
cp s-rack_amd/libsrack_hip_poison.so s-rack_amd/libsrack_hip.so
( SOAK_VT=16,48000 timeout 600 python tools/fuzz_soak_default.py 51666 51900 ) > $OUT/b1_poison_1s.log 2>&1; echo "== poison 1s rc=$?"; tail -6 $OUT/b1_poison_1s.log | cut -c1-250
( timeout 600 python tools/fuzz_soak_default.py 80000 82000 ) > $OUT/b1_poison_short.log 2>&1; echo "== poison short rc=$?"; tail -6 $OUT/b1_poison_short.log | cut -c1-250
( FUZZ_SPECIAL=1 timeout 900 python tools/fuzz_soak_default.py 80000 80300 ) > $OUT/b1_poison_special.log 2>&1; echo "== poison special rc=$?"; tail -6 $OUT/b1_poison_special.log | cut -c1-250
cp /tmp/_new.so s-rack_amd/libsrack_hip.so
( HORIZON_FLAGS=0,32,34,1,64,96 timeout 900 python tools/horizon.py $OUT/horizon_new.json ) > $OUT/b1_horizon.log 2>&1; echo "== horizon rc=$?"; cat $OUT/b1_horizon.log | cut -c1-230
( bash tools/ab.sh "--no-side-configs" /tmp/_new.so s-rack_amd/libsrack_hip_spec.so ) > $OUT/b1_ab_spec.log 2>&1; echo "== A/B speculative tile"; cat $OUT/b1_ab_spec.log | tail -8
cp /tmp/_new.so s-rack_amd/libsrack_hip.so
( timeout 900 python bench.py ) > $OUT/b1_bench.json 2> $OUT/b1_bench.err; echo "== bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/b1_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("cfg3 %.3f ms frac %.4f kernel %.4f" % (d["ms_per_step"], r["frac"], r["frac_kernel"]))
for k in sorted(r):
    if k.endswith("_ms_per_step"): print("  ", k, round(r[k],3), "frac_hbm", round(r.get(k.replace("_ms_per_step","_frac_hbm"),0),4))
PY
tail -3 $OUT/b1_bench.err | cut -c1-300
