#!/bin/bash
set -u
OUT=gpurun_out/r3f
mkdir -p $OUT
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("   %-28s ms/step %.3f  kernel %s x%d %.4f ms" % (sys.argv[2], d["ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
except Exception as e: print("   parse failed", sys.argv[1], e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
run() { name=$1; args=$2; shift; shift; env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --workload cfg4_b1024 $args > $OUT/$name.json 2> $OUT/$name.err; line $OUT/$name.json $name; }
run block_full "" A=1
run block_noframes "--no-frames" A=1
run block_nomix "--no-mix" A=1
run ring_noframes "--no-frames" SRACK_FM_BLOCK=0
run ring_nomix "--no-mix" SRACK_FM_BLOCK=0
