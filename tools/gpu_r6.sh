#!/bin/bash
# round 6's GPU passes, one parameterised script (run ON THE GPU BOX from the repo root):
#   tests [pytest args]          the -m gpu suite with its wall time
#   line                         the default bench line exactly as the driver runs it (python bench.py --steps 20 --warmup 5), its size and digest
#   bench <name> <bench args>    one bench line (5 steps, no CPU leg, no side configs)
#   ab "<bench args>" A.so B.so  alternate prebuilt libraries on this box, three rounds
#   prof <tag> "<bench args>"    rocprofv3 evidence for one workload (profiles/run_profile.sh)
set -u
OUT=gpurun_out/r6
mkdir -p $OUT
export TMPDIR=/tmp
digest() { python - "$1" <<'PY'
import json, sys
try:
    text = open(sys.argv[1]).read().strip().splitlines()[-1]
    d = json.loads(text)
    r = d["roofline"]
    print("   line %d bytes; ms/step %.3f  value %.4e  frac %.4f  frac_kernel %.4f  kernel %s x%d %.4f ms" % (len(text), d["ms_per_step"], d["value"], r.get("hbm", r)["frac"], r.get("hbm", r)["frac_kernel"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
    for k in sorted(r):
        if k.endswith("_ms_per_step"):
            w = k[:-len("_ms_per_step")]
            print("   %-18s %8.3f ms  frac %.4f  kernel %.4f" % (w, r[k], r.get(w + "_frac_hbm", 0), r.get(w + "_frac_hbm_kernel", 0)))
    c = d.get("cpu_baseline")
    if c:
        print("   cpu_baseline %.4e on %d cores (single thread %.4e)" % (c["value"], c["cores"], c["single_thread_value"]))
except Exception as e:
    print("   parse failed:", e)
PY
}
while [ $# -gt 0 ]; do
  case $1 in
    tests) shift; ( time timeout 1500 python -m pytest tests -m gpu -q -x ${1:-} ) > $OUT/pytest.log 2>&1; grep -E "passed|failed|error|^real" $OUT/pytest.log | tail -4; shift ;;
    line) shift; ( time python bench.py --steps 20 --warmup 5 ) > $OUT/default_line.json 2> $OUT/default_line.err; grep -E "^real" $OUT/default_line.err; digest $OUT/default_line.json; cp bench_detail.json $OUT/default_detail.json 2>/dev/null ;;
    bench) name=$2; args=$3; shift 3; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-side-configs $args > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; digest $OUT/$name.json ;;
    ab) args=$2; shift 2; libs=""
        while [ $# -gt 0 ] && [[ "$1" == *.so ]]; do libs="$libs $1"; shift; done
        cp s-rack_amd/libsrack_hip.so /tmp/keep.so
        for round in 1 2 3; do for lib in $libs; do
          cp $lib s-rack_amd/libsrack_hip.so
          timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --no-side-configs $args > $OUT/ab.json 2> $OUT/ab.err
          python - "$lib" $OUT/ab.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("   %-40s ms/step %.3f  kernel %.4f ms x%d" % (sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["launches_per_step"]))
except Exception as e:
    print("   %s parse failed: %s" % (sys.argv[1], e))
PY
        done; done
        cp /tmp/keep.so s-rack_amd/libsrack_hip.so ;;
    prof) tag=$2; args=$3; shift 3; bash profiles/run_profile.sh $tag "$args" > $OUT/prof_$tag.log 2>&1; tail -4 $OUT/prof_$tag.log
          mkdir -p $OUT/profiles; cp profiles/${tag}_* $OUT/profiles/ 2>/dev/null; rm -rf gpurun_out/prof_$tag/*/  # (the condensed files travel back; the raw databases do not)
          ;;
    *) echo "unknown pass $1"; shift ;;
  esac
done
