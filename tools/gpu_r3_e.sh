#!/bin/bash
# round 3: the time-parallel FM pair (buffer_size 256 ... 1024): parity, then timing against the ring kernel and over launch lengths
set -u
OUT=gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fm or cfg4 or p2 or ring or keep_state" ) > $OUT/pytest_fm.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest_fm.log | tail -12 | cut -c1-300
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("   %-28s ms/step %.3f  kernel %s x%d %.4f ms" % (sys.argv[2], d["ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
except Exception as e: print("   parse failed", sys.argv[1], e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --workload cfg4_b1024 > $OUT/$name.json 2> $OUT/$name.err; line $OUT/$name.json $name; }
for round in 1 2; do
  run ring SRACK_FM_BLOCK=0
  run block_seg SRACK_FM_BLOCK=1
  run block_16384 SRACK_FM_BLOCK_CHUNK=16384
  run block_4096 SRACK_FM_BLOCK_CHUNK=4096
done
