#!/bin/bash
# round 3, profile sets at HEAD (after tick sessions): every r03_* tag again + the ticked flagship
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3prof
( time bash tools/gpu_prof.sh ) 2>&1 | tail -40
echo "=== r03_tick (--block 1024)"
bash profiles/run_profile.sh r03_tick "--block 1024" > gpurun_out/r3prof/r03_tick.log 2>&1
grep -E "^(ok|FAIL) |summarize rc" gpurun_out/r3prof/r03_tick.log | cut -c1-220
mkdir -p gpurun_out/profiles && cp profiles/r03* gpurun_out/profiles/
rm -rf gpurun_out/prof_r03*/
