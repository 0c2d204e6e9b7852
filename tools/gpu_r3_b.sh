#!/bin/bash
# round 3, second GPU pass: the whole -m gpu suite; config 4 through the general path with and without the wave's vote; chaotic-seed scan
set -u
OUT=gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log | cut -c1-300
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("   %s: ms/step %.3f  kernel %s x%d %.4f ms" % (sys.argv[1].split('/')[-1], d["ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
except Exception as e: print("   parse failed", sys.argv[1], e)
PY
}
for round in 1 2; do
  SRACK_JIT_FM=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --workload cfg4 --flags 2 > $OUT/cfg4_special_nofm.json 2>$OUT/err; line $OUT/cfg4_special_nofm.json
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --workload cfg4 --flags 2 > $OUT/cfg4_special_fm.json 2>$OUT/err; line $OUT/cfg4_special_fm.json
  SRACK_CHUNK_MAX=2048 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --workload cfg4 --flags 2 > $OUT/cfg4_special_fm_2048.json 2>$OUT/err; line $OUT/cfg4_special_fm_2048.json
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --workload cfg4 > $OUT/cfg4.json 2>$OUT/err; line $OUT/cfg4.json
done
timeout 600 python tools/fuzz_soak_default.py 1400 1500 > $OUT/soak_default_1400.log 2>&1; tail -8 $OUT/soak_default_1400.log
timeout 600 python tools/fuzz_soak_default.py 700 760 > $OUT/soak_default_700.log 2>&1; tail -8 $OUT/soak_default_700.log
