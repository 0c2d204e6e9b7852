#!/bin/bash
# the one-cycle-per-sample check of the default increment (modules.hip.h, osc_delta_fast): library before (r4f) against after (r4g) on ONE
# box — P3 and config 4 (three alternating rounds), then the random-patch survey's first 30 seeds
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
cp s-rack_amd/libsrack_hip.so /tmp/head.so
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.3f kernel_ms %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]), d["config"]["program"][-50:])
PY
}
for round in 1 2 3; do for v in r4f r4g; do
  cp tools/ab_libs/$v.so s-rack_amd/libsrack_hip.so
  for w in p3 cfg4 cfg2; do
    timeout 600 python bench.py --no-cpu --no-side-configs --workload $w --steps 10 --warmup 2 > $OUT/delta_${v}_${w}_$round.json 2>$OUT/delta.err
    line $OUT/delta_${v}_${w}_$round.json "$v $w"
  done
done; done
for v in r4f r4g; do
  cp tools/ab_libs/$v.so s-rack_amd/libsrack_hip.so
  python tools/patch_survey.py 0 30 262144 6000 > $OUT/survey_delta_$v.txt 2>&1
done
cp /tmp/head.so s-rack_amd/libsrack_hip.so
python tools/survey_compare.py $OUT/survey_delta_r4f.txt $OUT/survey_delta_r4g.txt | tail -45
