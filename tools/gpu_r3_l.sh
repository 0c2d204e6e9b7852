#!/bin/bash
# round 3: what a host that ticks block by block pays (one srack_render call per buffer_size samples, main.rs:59-63) against one call per second
set -u
OUT=gpurun_out/r3l
mkdir -p $OUT
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("   %-28s ms/step %.3f  host enqueue %.3f  kernel %s x%d %.4f ms" % (sys.argv[2], d["ms_per_step"], d["host_enqueue_ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
except Exception as e: print("   parse failed", sys.argv[1], e)
PY
}
for w in cfg3 cfg4 cfg4_b1024 cfg2; do
  for b in 0 4096 1024 256; do
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-side-configs --workload $w --block $b > $OUT/${w}_b$b.json 2>$OUT/err || tail -3 $OUT/err
    line $OUT/${w}_b$b.json "$w block=$b"
  done
done
for b in 0 1024; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-side-configs --workload p3 --no-frames --block $b > $OUT/p3_b$b.json 2>$OUT/err || tail -3 $OUT/err
  line $OUT/p3_b$b.json "p3 (mix only) block=$b"
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-side-configs --flags 2 --block $b > $OUT/special_b$b.json 2>$OUT/err || tail -3 $OUT/err
  line $OUT/special_b$b.json "cfg3 general path block=$b"
done
