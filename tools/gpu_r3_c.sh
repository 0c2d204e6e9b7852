#!/bin/bash
# round 3, third GPU pass: suite, default bench line, FM pair special-vs-fused A/B, the profile sets
set -u
OUT=gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log | cut -c1-300
( time timeout 300 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c/bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("default: ms/step %.3f frac %.4f frac_kernel %.4f" % (d["ms_per_step"], r["frac"], r["frac_kernel"]), {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k.startswith("cfg")})
print("  cfg2:", d["configs"]["cfg2"]["program"][-120:]); print("  cfg4:", d["configs"]["cfg4"]["program"][-120:])
PY
tail -3 $OUT/bench_default.err
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("   %s: ms/step %.3f  kernel %s x%d %.4f ms" % (sys.argv[1].split('/')[-1], d["ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
except Exception as e: print("   parse failed", sys.argv[1], e)
PY
}
for round in 1 2 3; do
  SRACK_FM_FUSED=1 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --workload cfg4 > $OUT/cfg4_fused.json 2>$OUT/err; line $OUT/cfg4_fused.json
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --workload cfg4 > $OUT/cfg4_special.json 2>$OUT/err; line $OUT/cfg4_special.json
done
bash tools/gpu_prof.sh
