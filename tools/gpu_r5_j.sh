#!/bin/bash
# Round 5, last pass at HEAD: config 4's profiles again (its kernel changed after pass E: pow's table in LDS, the rounding check), the curves, the default line
set -u
OUT=gpurun_out/r5; mkdir -p $OUT gpurun_out/profiles
prof() { tag=$1; shift; bash profiles/run_profile.sh "$tag" "$*" > $OUT/prof_$tag.log 2>&1; grep -E "^(ok|FAIL) |summarize rc" $OUT/prof_$tag.log | cut -c1-200 | tail -4; cp profiles/${tag}_* gpurun_out/profiles/ 2>/dev/null; rm -rf gpurun_out/prof_$tag; }
prof r05_cfg4 --workload cfg4
prof r05_cfg4_b1024 --workload cfg4_b1024
( HORIZON_FLAGS=0,32,34,1,64 timeout 900 python tools/horizon.py gpurun_out/profiles/r05_horizon.json ) > $OUT/j_horizon.log 2>&1; echo "== horizon rc=$?"; grep -c worst $OUT/j_horizon.log
( timeout 900 python bench.py ) > gpurun_out/profiles/r05_default_line.json 2> $OUT/j_bench.err; echo "== bench rc=$?"; cut -c1-300 gpurun_out/profiles/r05_default_line.json
