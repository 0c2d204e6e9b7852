#!/bin/bash
# round 4's GPU passes, one parameterised script (run ON THE GPU BOX from the repo root through gpurun):
#   bash tools/gpu_r4.sh <pass> [args]
# passes:  tests [pytest -k expr]    the -m gpu suite (or a selection)
#          bench <name> <bench args> one bench line -> gpurun_out/r4/<name>.json (+ a one-line digest)
#          ab <VAR> "<v1 v2>" <bench args>    an environment knob, three alternating rounds on this box
set -u
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
digest() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]; h = r.get("hbm", r)
    print("   ms/step %.3f  value %.3e  frac %.3f  frac_kernel %.3f  kernel %s x%d %.3f ms  %s" % (d["ms_per_step"], d["value"], h["frac"], h["frac_kernel"], r["kernel"], r["launches_per_step"], r["kernel_ms"], d["config"]["program"][-60:]))
except Exception as e:
    print("   parse failed", e)
PY
}
case ${1:-} in
  tests) shift; ( time timeout 2400 python -m pytest tests -m gpu -q -x ${1:+-k "$1"} ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log ;;
  bench) name=$2; shift 2; timeout 600 python bench.py --no-cpu --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; digest $OUT/$name.json ;;
  ab) var=$2; vals=$3; shift 3
      for round in 1 2 3; do for v in $vals; do
        env $var=$v timeout 600 python bench.py --no-cpu --no-side-configs "$@" > $OUT/ab_${var}_${v}_$round.json 2> $OUT/ab.err; echo "$var=$v round $round"; digest $OUT/ab_${var}_${v}_$round.json
      done; done ;;
  power) # power <tag> <bench args>: package power and sclk (rocm-smi) during a long run -> gpurun_out/profiles/r04_power_<tag>.log
      tag=$2; shift 2; mkdir -p gpurun_out/profiles
      bash tools/power_probe.sh "--no-side-configs $*" > gpurun_out/profiles/r04_power_$tag.log 2>&1; tail -4 gpurun_out/profiles/r04_power_$tag.log ;;
  prof) # prof <tag> [bench args]: the rocprofv3 evidence (profiles/run_profile.sh) -> gpurun_out/profiles/<tag>_*
      tag=$2; shift 2; mkdir -p gpurun_out/profiles
      bash profiles/run_profile.sh "$tag" "$*" > $OUT/prof_$tag.log 2>&1; grep -E "^(ok|FAIL) |summarize rc" $OUT/prof_$tag.log | cut -c1-220
      cp profiles/${tag}_* gpurun_out/profiles/; rm -rf gpurun_out/prof_$tag ;;
  *) echo "usage: gpu_r4.sh tests|bench|ab|power|prof ..." ;;
esac
