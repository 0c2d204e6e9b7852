#!/bin/bash
# round 3: tick sessions — their tests, the tests the call-length rule of the time-parallel FM pair touched, then the block-by-block timings again
set -u
OUT=gpurun_out/r3m
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_tick.py tests/test_srk.py tests/test_gpu_parity.py -m gpu -q -x -k "tick or fm_pair or feedback_ring or cfg4 or keep_state or split or contin" ) > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log | cut -c1-400
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("   %-28s ms/step %.3f  host enqueue %.3f  kernel %s x%d %.4f ms" % (sys.argv[2], d["ms_per_step"], d["host_enqueue_ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
except Exception as e: print("   parse failed", sys.argv[1], e)
PY
}
for t in 1 0; do
  echo "SRACK_TICK=$t"
  for spec in "cfg3|" "cfg2|" "cfg4_b1024|" "p3|--no-frames" "cfg3|--flags 2"; do
    IFS='|' read -r w extra <<< "$spec"
    for b in 1024 256; do
      SRACK_TICK=$t timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-side-configs --workload $w $extra --block $b > $OUT/${w}_t${t}_b$b.json 2>$OUT/err || tail -3 $OUT/err
      line $OUT/${w}_t${t}_b$b.json "$w $extra block=$b"
    done
  done
done
