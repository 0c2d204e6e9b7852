#!/bin/bash
# quick pass: parity suite + selected bench lines (arguments: names of bench variants to run)
set -u
OUT=gpurun_out/b
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} ) > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -3
run() { name=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("   ms/step %.2f  value %.3e  frac %.3f  frac_kernel %.3f  kernel %s x%d %.3f ms" % (d["ms_per_step"], d["value"], r.get("hbm",r)["frac"], r.get("hbm",r)["frac_kernel"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
except Exception as e:
    print("   parse failed", e)
PY
}
for v in "$@"; do
  case $v in
    cfg3) run cfg3 ;;
    exact) run cfg3_exact --flags 1 ;;
    interp) run cfg3_interp --flags 2 ;;
    interp_exact) run cfg3_interp_exact --flags 3 ;;
    nohoist) run cfg3_nohoist --flags 4 ;;
    nohoist_exact) run cfg3_nohoist_exact --flags 5 ;;
    cfg2) run cfg2 --workload cfg2 ;;
    cfg4) run cfg4 --workload cfg4 ;;
    cfg4b) run cfg4_b1024 --workload cfg4_b1024 ;;
    p3) run p3 --workload p3 ;;
    p3_dist) run p3_dist --workload p3 --force-dist ;;
    p3_interp) run p3_interp --workload p3 --flags 2 ;;
  esac
done
for v in "$@"; do
  case $v in
    special_nohoist) run cfg3_special_nohoist --flags 6 ;;
    p3_special) run p3_special --workload p3 --flags 2 ;;
    cfg4_special) run cfg4_special --workload cfg4 --flags 2 ;;
    cfg4b_special) run cfg4b_special --workload cfg4_b1024 --flags 2 ;;
    cfg2_special) run cfg2_special --workload cfg2 --flags 34 ;;
    interp_only) run cfg3_interp_only --flags 18 ;;
    p4) run p4 --workload p4 ;;
    p4_interp) run p4_interp --workload p4 --flags 16 ;;
  esac
done
for v in "$@"; do
  case $v in
    cfg4_exact) run cfg4_exact --workload cfg4 --flags 1 --steps 2 --warmup 1 ;;
    cfg4_special_exact) run cfg4_special_exact --workload cfg4 --flags 3 --steps 2 --warmup 1 ;;
    p3_exact) run p3_exact --workload p3 --flags 1 --steps 2 --warmup 1 ;;
  esac
done
