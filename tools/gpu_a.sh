#!/bin/bash
# Round-2 GPU pass A: parity suite, the bench lines of every workload / mode, with and without the RCCL communicator.
set -u
OUT=gpurun_out/a
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
run() { name=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("   ms/step %.2f  value %.3e  frac %.3f  frac_kernel %.3f  kernel %s x%d %.3f ms ranks_seen %s" % (d["ms_per_step"], d["value"], r.get("hbm",r)["frac"], r.get("hbm",r)["frac_kernel"], r["kernel"], r["launches_per_step"], r["kernel_ms"], d.get("ranks_seen")))
except Exception as e:
    print("   parse failed", e)
PY
}
run cfg3
run cfg3_dist --force-dist --no-cpu
run cfg3_exact --flags 1 --no-cpu
run cfg3_interp --flags 2 --no-cpu
run cfg3_interp_exact --flags 3 --no-cpu
run cfg3_nohoist --flags 4 --no-cpu
run cfg2 --workload cfg2
run cfg4 --workload cfg4
run cfg4_b1024 --workload cfg4_b1024 --no-cpu
run p3 --workload p3
run p3_dist --workload p3 --force-dist
run p3_interp --workload p3 --flags 2
run p3_interp_dist --workload p3 --flags 2 --force-dist
SRACK_CTL_HIGH_PRIO=0 run p3_dist_noprio --workload p3 --force-dist
./tools/rtc_probe > $OUT/rtc_probe.log 2>&1; echo "rtc_probe rc=$?"; tail -2 $OUT/rtc_probe.log
