#!/bin/bash
# round 3, first GPU pass: whole -m gpu suite, the default bench line (with configs 2 and 4 on it), the general path on config 4, one profile set
set -u
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fuzz.py::test_random_patch_default_modes_within_tolerance ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
( time timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k default_modes ) > $OUT/pytest_default.log 2>&1
grep -E "^FAILED|passed|failed" $OUT/pytest_default.log | cut -c1-400 | tail -40
( time timeout 300 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
for spec in "cfg4_special|--workload cfg4 --flags 2" "cfg4|--workload cfg4" "cfg4b|--workload cfg4_b1024" "cfg4b_special|--workload cfg4_b1024 --flags 2"; do
  IFS='|' read -r name args <<< "$spec"
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu $args > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$name ms/step %.3f  kernel %s x%d %.3f ms" % (d["ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
PY
done
bash profiles/run_profile.sh r03 "" " " > $OUT/prof_r03.log 2>&1
tail -30 $OUT/prof_r03.log
mkdir -p gpurun_out/profiles && cp profiles/r03* gpurun_out/profiles/
rm -rf gpurun_out/prof_r03/*/  # raw databases stay on the box
