#!/bin/bash
# round 3, fourth GPU pass: the MFMA mix-down experiment (A/B on one box, three alternating rounds) + its parity, and two profile sets again
set -u
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/ab.sh "--no-side-configs" tools/ab_libs/base.so tools/ab_libs/mfma.so 2>&1 | tee $OUT/ab_mfma.log
# parity of the variant: the flagship's tests (mix vs f64 sums, golden voices) with the MFMA library in place
cp s-rack_amd/libsrack_hip.so /tmp/_keep.so; cp tools/ab_libs/mfma.so s-rack_amd/libsrack_hip.so
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -q -k "cfg3 or p1_voices or mix or one_rank or cfg1" ) > $OUT/pytest_mfma.log 2>&1; tail -4 $OUT/pytest_mfma.log | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-side-configs --no-frames > $OUT/mfma_noframes.json 2>$OUT/err
cp /tmp/_keep.so s-rack_amd/libsrack_hip.so
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-side-configs --no-frames > $OUT/base_noframes.json 2>$OUT/err
python - <<'PY'
import json
for n in ("base_noframes","mfma_noframes"):
    d=json.loads(open(f"gpurun_out/r3d/{n}.json").read().strip().splitlines()[-1]); print(n, round(d["ms_per_step"],3))
PY
for spec in "r03_cfg2|--workload cfg2" "r03_cfg4|--workload cfg4"; do
  IFS='|' read -r tag args <<< "$spec"
  bash profiles/run_profile.sh "$tag" "$args" > gpurun_out/r3prof/$tag.log 2>&1
  grep -E "^(ok|FAIL) |summarize rc" gpurun_out/r3prof/$tag.log | cut -c1-220
done
mkdir -p gpurun_out/profiles && cp profiles/r03_cfg2* profiles/r03_cfg4_* gpurun_out/profiles/
rm -rf gpurun_out/prof_r03*/
